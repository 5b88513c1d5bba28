"""Independent NumPy/SciPy restatement of the reference SfT solve -- TEST INFRASTRUCTURE ONLY.

Purpose: cross-check oracle/sft_oracle.c with a second, structurally different
implementation (a global sparse Jacobian and H = J^T W J instead of per-edge block
accumulation; scipy rotations instead of hand-written quaternions) and emit the
golden fixtures under tests/golden/ (see tests/golden/make_golden.py).

Follows the same reference lines as sft_oracle.c: sft_types.h:75-411 (residuals and
the reference's own Jacobians, including the per-node-depth approximation of
EdgeNodesCamera), DefOptimizer.cc:251-513 (graph and weights),
optimization_algorithm_levenberg.cpp:61-189 (controller), robust_kernel_impl.cpp:78-91.
PARITY UNPINNED (no reference vectors exist; g2o cannot be built here).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
from scipy.spatial.transform import Rotation


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(u):
    """se3quat.h:223-257; returns (R, t)."""
    om, up = u[:3], u[3:]
    th = np.linalg.norm(om)
    Om = _skew(om)
    if th < 0.00001:
        R = np.eye(3) + Om + Om @ Om
        V = R
    else:
        Om2 = Om @ Om
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th**2 * Om2
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * Om + (th - np.sin(th)) / th**3 * Om2
    # the reference converts R to a unit quaternion (renormalising it)
    R = Rotation.from_matrix(R).as_matrix()
    return R, V @ up


class Graph:
    def __init__(self, tc, Tcw, K, n_frame, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz, reg_lap, reg_inex, reg_temp, layers=1):
        self.tc = tc
        n = tc.n
        self.n = n
        self.K = np.asarray(K, float)
        T = np.asarray(Tcw, np.float32).astype(float)
        self.R = Rotation.from_matrix(T[:3, :3]).as_matrix()
        self.t = T[:3, 3].copy()
        self.xyz = np.array(xyz, float)
        self.obs_nodes = np.asarray(obs_nodes)
        self.obs_bary = np.asarray(obs_bary, float)
        self.obs_uv = np.asarray(obs_uv, float)
        self.w_obs = np.asarray(obs_invsig2, float) / float(n_frame)
        d = float(np.float32(np.sqrt(5.991)))
        self.delta, self.dsqr = d, d * d
        viewed = np.zeros(n, bool)
        viewed[self.obs_nodes.ravel()] = True
        opt = viewed.copy()
        if layers >= 1:
            for i in np.nonzero(viewed)[0]:
                opt[tc.nbr_idx[tc.nbr_ptr[i]:tc.nbr_ptr[i + 1]]] = True
        self.viewed, self.opt = viewed, opt
        self.ref_nodes = np.nonzero(viewed)[0]
        self.w_ref = reg_temp / tc.median_L**2
        n_opt = int(opt.sum())
        # curvature edges: (centre, incident edge) pairs
        self.curv = [(i, e) for i in range(n) if opt[i] and not tc.boundary[i] for e in tc.inc_edge[tc.inc_ptr[i]:tc.inc_ptr[i + 1]]]
        self.w_curv = reg_lap / n_opt
        eact = np.zeros(tc.E, bool)
        for i in np.nonzero(opt)[0]:
            eact[tc.inc_edge[tc.inc_ptr[i]:tc.inc_ptr[i + 1]]] = True
        self.stretch = np.nonzero(eact)[0]
        self.w_str = reg_inex / len(self.stretch)
        # unknown layout: camera 0..5, then active nodes ascending
        self.col = -np.ones(n, int)
        self.col[opt] = 6 + 3 * np.arange(n_opt)
        self.D = 6 + 3 * n_opt

    # -- residual vector in the reference's edge order, with per-residual weights ----------
    def residuals(self):
        tc = self.tc
        fx, fy, cx, cy = self.K
        pw = (self.obs_bary[:, :, None] * self.xyz[self.obs_nodes]).sum(1)
        pc = pw @ self.R.T + self.t
        e_obs = self.obs_uv - np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1)
        e_ref = self.xyz[self.ref_nodes] - tc.xyz0[self.ref_nodes]
        e_curv = np.zeros(len(self.curv))
        self._mc = {}
        for k, (i, e) in enumerate(self.curv):
            if i not in self._mc:
                nb = tc.nbr_idx[tc.nbr_ptr[i]:tc.nbr_ptr[i + 1]]
                w = tc.nbr_w[tc.nbr_ptr[i]:tc.nbr_ptr[i + 1]]
                mc = self.xyz[i] - (w[:, None] * self.xyz[nb]).sum(0) / w.sum()
                self._mc[i] = (mc, np.linalg.norm(mc), nb, w)
            mc, nrm, _, _ = self._mc[i]
            e_curv[k] = (nrm - tc.k0[i]) / tc.edge_L0[e]
        a, b = tc.edge_nodes[self.stretch, 0], tc.edge_nodes[self.stretch, 1]
        e_str = np.linalg.norm(self.xyz[a] - self.xyz[b], axis=1) / tc.edge_L0[self.stretch] - 1.0
        return e_obs, e_ref, e_curv, e_str

    def chi2_parts(self, res):
        e_obs, e_ref, e_curv, e_str = res
        c_obs = self.w_obs * (e_obs**2).sum(1)
        return c_obs, self.w_ref * (e_ref**2).sum(1), self.w_curv * e_curv**2, self.w_str * e_str**2

    def robust_chi2(self, res):
        c_obs, c_ref, c_curv, c_str = self.chi2_parts(res)
        rho = np.where(c_obs <= self.dsqr, c_obs, 2 * np.sqrt(np.maximum(c_obs, 1e-300)) * self.delta - self.dsqr)
        return rho.sum() + c_ref.sum() + c_curv.sum() + c_str.sum()

    def system(self, res):
        """Sparse J (rows = residual scalars), weights, then dense H and b."""
        tc = self.tc
        e_obs, e_ref, e_curv, e_str = res
        fx, fy, cx, cy = self.K
        rows, cols, vals, wts, errs = [], [], [], [], []
        r = 0
        c_obs = self.w_obs * (e_obs**2).sum(1)
        rho1 = np.where(c_obs <= self.dsqr, 1.0, self.delta / np.sqrt(np.maximum(c_obs, 1e-300)))
        for m in range(len(e_obs)):
            nd = self.obs_nodes[m]
            pcs = self.xyz[nd] @ self.R.T + self.t
            x, y, z = (self.obs_bary[m][:, None] * pcs).sum(0)
            Jc = np.array([[x * y / z**2 * fx, -(1 + x * x / z**2) * fx, y / z * fx, -1 / z * fx, 0, x / z**2 * fx],
                           [(1 + y * y / z**2) * fy, -x * y / z**2 * fy, -x / z * fy, 0, -1 / z * fy, y / z**2 * fy]])
            for rr in range(2):
                for cc in range(6):
                    rows.append(r + rr); cols.append(cc); vals.append(Jc[rr, cc])
            for s in range(3):
                xs, ys, zs = pcs[s]
                tmp = np.array([[fx, 0, -xs / zs * fx], [0, fy, -ys / zs * fy]])
                Jn = (-1.0 / zs) * tmp @ self.R * self.obs_bary[m, s]
                c0 = self.col[nd[s]]
                if c0 >= 0:
                    for rr in range(2):
                        for cc in range(3):
                            rows.append(r + rr); cols.append(c0 + cc); vals.append(Jn[rr, cc])
            wts += [rho1[m] * self.w_obs[m]] * 2
            errs += list(e_obs[m])
            r += 2
        for k, i in enumerate(self.ref_nodes):
            c0 = self.col[i]
            for cc in range(3):
                rows.append(r + cc); cols.append(c0 + cc); vals.append(1.0)
            wts += [self.w_ref] * 3
            errs += list(e_ref[k])
            r += 3
        for k, (i, e) in enumerate(self.curv):
            mc, nrm, nb, w = self._mc[i]
            L = tc.edge_L0[e]
            if nrm >= 1e-15:
                g = mc / (nrm * L)
                for cc in range(3):
                    rows.append(r); cols.append(self.col[i] + cc); vals.append(g[cc])
                for j, wj in zip(nb, w):
                    if self.col[j] >= 0:
                        for cc in range(3):
                            rows.append(r); cols.append(self.col[j] + cc); vals.append(-(wj / w.sum()) * g[cc])
            wts.append(self.w_curv); errs.append(e_curv[k]); r += 1
        for k, e in enumerate(self.stretch):
            a, b = tc.edge_nodes[e]
            d = self.xyz[a] - self.xyz[b]
            g = d / (np.linalg.norm(d) * tc.edge_L0[e])
            for cc in range(3):
                if self.col[a] >= 0:
                    rows.append(r); cols.append(self.col[a] + cc); vals.append(g[cc])
                if self.col[b] >= 0:
                    rows.append(r); cols.append(self.col[b] + cc); vals.append(-g[cc])
            wts.append(self.w_str); errs.append(e_str[k]); r += 1
        J = sp.csr_matrix((vals, (rows, cols)), shape=(r, self.D))
        W = sp.diags(np.asarray(wts))
        H = (J.T @ W @ J).toarray()
        b = -(J.T @ (np.asarray(wts) * np.asarray(errs)))
        return H, b

    def apply(self, x):
        dR, dt = se3_exp(x[:6])
        self.t = dR @ self.t + dt
        self.R = Rotation.from_matrix(dR @ self.R).as_matrix()
        act = np.nonzero(self.opt)[0]
        self.xyz[act] += x[6:].reshape(-1, 3)


def solve(tc, Tcw, K, n_frame, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz, reg_lap, reg_inex, reg_temp, layers=1, max_iters=50):
    g = Graph(tc, Tcw, K, n_frame, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz, reg_lap, reg_inex, reg_temp, layers)
    lam, ni, n_bad = -1.0, 2.0, 0
    trace = []
    res = None
    for it in range(max_iters):
        res = g.residuals()
        cur = g.robust_chi2(res)
        ini = cur
        H, b = g.system(res)
        if it == 0:
            lam, ni, n_bad = 1e-5 * np.abs(np.diag(H)).max(), 2.0, 0
        lam0 = lam
        rho, q = 0.0, 0
        accepted = 0
        while True:
            bak = (g.R.copy(), g.t.copy(), g.xyz.copy())
            try:
                cf = sla.cho_factor(H + lam * np.eye(g.D), lower=True)
                x = sla.cho_solve(cf, b)
                ok = True
            except sla.LinAlgError:
                ok = False
                x = np.zeros(g.D)
            g.apply(x)
            res = g.residuals()
            tmp = g.robust_chi2(res) if ok else np.finfo(float).max
            rho = (cur - tmp) / (float(x @ (lam * x + b)) + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                cur = tmp
                accepted = 1
            else:
                lam *= ni
                ni *= 2
                g.R, g.t, g.xyz = bak
            q += 1
            if not (rho < 0 and q < 10):
                break
        trace.append([ini, lam0, q, cur, lam, rho, accepted, 0])
        if q == 10 or rho == 0:
            break
        n_bad = n_bad + 1 if (ini - cur) * 1e3 < ini else 0
        if n_bad >= 3:
            break
    c_obs = g.chi2_parts(res)[0]  # last evaluated errors, like the reference
    outlier = (c_obs.astype(np.float32) > 5.991)
    e_fin = g.residuals()[0]
    rep = np.sqrt((e_fin[~outlier] ** 2).sum(1)).sum() / max(int((~outlier).sum()), 1)
    q = Rotation.from_matrix(g.R).as_quat()
    if q[3] < 0:
        q = -q
    return dict(pose7=np.concatenate([g.t, q]), xyz=g.xyz, chi2_obs=c_obs, outlier=outlier, rep_error=rep,
                iters=len(trace), trace=np.asarray(trace), ret=int((~outlier).sum()))
