"""Seeded synthetic SfT inputs.

The reference has no dataset that can be loaded here (OpenCV + Mandala/Hamlyn
sequences absent), so tests and bench use this generator: a regular-grid template
at z~1 in front of a 640x480 pinhole camera, a smooth ground-truth deformation,
a small ground-truth camera motion and barycentric-embedded "ORB matches" with
pixel noise and outliers.  Values that are float32 in the reference (keypoints,
invSigma2, barycentrics, the 4x4 pose) are rounded through float32 here so both
the oracle and the HIP path see exactly what DefPoseOptimization would see
(Modules/Tracking/DefOptimizer.cc:293-361).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# Regularisers of the Mandala sequences (reference yaml: Regularizer.laplacian/Inextensibility/temporal).
REG_LAP = 700.0
REG_INEX = 12000.0
REG_TEMP = 0.05
N_FRAME_KEYPOINTS = 1200
CAMERA_K = (500.0, 500.0, 320.0, 240.0)  # fx, fy, cx, cy for a 640x480 frame

CONFIGS = {
    # name: (rows, cols, matches)
    "smoke": (10, 10, 300),
    "REF": (10, 10, 450),   # the reference's own default size: 10 x 10 template (TriangularMesh.cc:63-64), 1200 features per frame of which a few hundred match
    "C2": (25, 20, 1000),   # 500-node template, 1000 matches (BASELINE.json configs[1])
    "C5": (50, 40, 4000),   # 2000-node template, 4000 matches (configs[4], per problem)
    "W12": (8, 30, 500),    # small wide-band cases for oracle-sized parity runs: half-bandwidth 182 (12 sub-diagonal tiles) ...
    "W16": (6, 41, 500),    # ... and 248 (16 tiles, the C5 band) with 246 nodes
    "B272": (5, 45, 400),   # half-bandwidth 272 > 256: the row-major band solver (general fallback)
}


@dataclass
class GridTemplate:
    rows: int
    cols: int
    xyz0: np.ndarray            # (n,3) float64 rest shape
    facets: np.ndarray          # (F,3) int32, as emitted by the regular triangulation (unsorted)

    @property
    def n(self) -> int:
        return self.rows * self.cols


@dataclass
class SftFrame:
    Tcw: np.ndarray             # (4,4) float32 initial pose (previous frame's pose)
    K: np.ndarray               # (4,) float64 fx,fy,cx,cy
    n_frame: int                # number of keypoints in the frame (scales the information)
    obs_facet: np.ndarray       # (M,) int32
    obs_nodes: np.ndarray       # (M,3) int32 ascending node ids of the facet
    obs_bary: np.ndarray        # (M,3) float64 (float32-valued)
    obs_uv: np.ndarray          # (M,2) float64 (float32-valued)
    obs_invsig2: np.ndarray     # (M,) float64 (float32-valued)
    xyz: np.ndarray             # (n,3) float64 current node estimates
    gt_xyz: np.ndarray = field(default=None)
    gt_Tcw: np.ndarray = field(default=None)
    is_outlier_gt: np.ndarray = field(default=None)


def regular_triangulation(rows: int, cols: int) -> np.ndarray:
    """Facet list of the reference's regular mesh (Modules/Template/TriangularMesh.cc:92-107),
    generalised to rows x cols with node id = col + cols*row."""
    f = []
    for j in range(rows - 1):
        for i in range(cols - 1):
            f.append((i + cols * j, i + cols * j + 1, cols * (j + 1) + i))
            f.append((i + cols * j + 1, cols * (j + 1) + i, cols * (j + 1) + i + 1))
    return np.asarray(f, dtype=np.int32)


def make_grid_template(rows: int, cols: int, seed: int = 1234, z0: float = 1.0) -> GridTemplate:
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = CAMERA_K
    # cover 80% of the frustum at depth z0
    xs = np.linspace(-0.8 * cx / fx, 0.8 * (640 - cx) / fx, cols) * z0
    ys = np.linspace(-0.8 * cy / fy, 0.8 * (480 - cy) / fy, rows) * z0
    X, Y = np.meshgrid(xs, ys)  # row-major: node id = col + cols*row
    ph = rng.uniform(0, 2 * np.pi, size=2)
    # low-frequency bump, amplitude 0.02, so the rest mean curvature is non-zero
    Z = z0 + 0.02 * np.sin(2 * np.pi * X / (xs[-1] - xs[0]) + ph[0]) * np.cos(2 * np.pi * Y / (ys[-1] - ys[0]) + ph[1])
    xyz0 = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    # the reference builds nodes from float32 vertices (TriangularMesh.cc:109-123)
    xyz0 = xyz0.astype(np.float32).astype(np.float64)
    return GridTemplate(rows, cols, xyz0, regular_triangulation(rows, cols))


def _rodrigues(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def make_frame(tmpl: GridTemplate, n_matches: int, problem_id: int = 0, *,
               bend: float = 0.05, rot_deg: float = 3.0, trans: float = 0.02,
               noise_px: float = 0.5, outlier_frac: float = 0.05,
               n_frame: int = N_FRAME_KEYPOINTS, init_xyz: np.ndarray | None = None,
               init_Tcw: np.ndarray | None = None, phase: float = 0.0, gt_pose: tuple | None = None) -> SftFrame:
    """gt_pose = (rotation vector, translation) fixes the ground-truth camera (sequences: a smooth trajectory) instead of
    drawing it from the problem's random stream; the stream is consumed identically either way."""
    rng = np.random.default_rng(42 + problem_id)
    fx, fy, cx, cy = CAMERA_K
    xyz0 = tmpl.xyz0
    width = xyz0[:, 0].max() - xyz0[:, 0].min()
    # smooth, roughly isometric bend: sinusoid with wavelength = mesh width
    gt = xyz0.copy()
    gt[:, 2] += bend * np.sin(2 * np.pi * (xyz0[:, 0] - xyz0[:, 0].min()) / width + phase + 0.3 * problem_id)
    # ground-truth camera: small SE3 offset
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rot_deg) * rng.uniform(0.3, 1.0)
    R = _rodrigues(axis * ang)
    t = rng.uniform(-trans, trans, size=3)
    if gt_pose is not None:
        R = _rodrigues(np.asarray(gt_pose[0], dtype=np.float64))
        t = np.asarray(gt_pose[1], dtype=np.float64)
    gtT = np.eye(4)
    gtT[:3, :3] = R
    gtT[:3, 3] = t

    F = tmpl.facets.shape[0]
    fac = rng.integers(0, F, size=n_matches).astype(np.int32)
    bary = rng.dirichlet((1.0, 1.0, 1.0), size=n_matches).astype(np.float32).astype(np.float64)
    nodes = np.sort(tmpl.facets[fac], axis=1).astype(np.int32)  # std::set<Node*> order
    pw = (bary[:, :, None] * gt[nodes]).sum(axis=1)
    pc = pw @ R.T + t
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], axis=1)
    uv += rng.normal(scale=noise_px, size=uv.shape)
    is_out = rng.uniform(size=n_matches) < outlier_frac
    uv_out = np.stack([rng.uniform(0, 640, size=n_matches), rng.uniform(0, 480, size=n_matches)], axis=1)
    uv = np.where(is_out[:, None], uv_out, uv)
    uv = uv.astype(np.float32).astype(np.float64)
    octave = rng.integers(0, 6, size=n_matches)
    invsig2 = (1.2 ** (-2.0 * octave)).astype(np.float32).astype(np.float64)

    Tcw = np.eye(4, dtype=np.float32) if init_Tcw is None else np.asarray(init_Tcw, dtype=np.float32)
    xyz = xyz0.copy() if init_xyz is None else np.asarray(init_xyz, dtype=np.float64).copy()
    return SftFrame(Tcw=Tcw, K=np.asarray(CAMERA_K, dtype=np.float64), n_frame=n_frame,
                    obs_facet=fac, obs_nodes=nodes, obs_bary=bary, obs_uv=uv, obs_invsig2=invsig2,
                    xyz=xyz, gt_xyz=gt, gt_Tcw=gtT, is_outlier_gt=is_out)


# ---------------------------------------------------------------------------------------------------------
# SEQ100 (the runnable substitute of BASELINE.json configs[0] / configs[2], the Mandala sequences that
# need OpenCV + datasets): a sequence of frames with temporally smooth deformation and camera motion.  Frame k is
# tracked from the result of frame k-1 (DefTracking.cc:350: each frame starts at the previous pose / mesh; the pose
# takes the reference's float32 round trip through cv::Mat).
# ---------------------------------------------------------------------------------------------------------
SEQ100 = dict(config="C2", n_frames=100, seq_id=0)


def sequence_gt_pose(k: int, n_frames: int, rot_deg: float = 3.0, trans: float = 0.02):
    """Ground-truth camera of frame k: a closed smooth loop (rotation <= rot_deg about a fixed axis, translation on a
    small ellipse), so consecutive frames differ by ~2*pi/n_frames of it."""
    a = 2.0 * np.pi * k / n_frames
    axis = np.array([0.4, 0.8, 0.2]) / np.linalg.norm([0.4, 0.8, 0.2])
    rotvec = axis * np.deg2rad(rot_deg) * np.sin(a)
    t = trans * np.array([np.sin(a), 0.5 * (1.0 - np.cos(a)), 0.3 * np.sin(2.0 * a)])
    return rotvec, t


def make_sequence_frame(tmpl: GridTemplate, n_matches: int, k: int, n_frames: int = 100, seq_id: int = 0,
                        init_xyz: np.ndarray | None = None, init_Tcw: np.ndarray | None = None) -> SftFrame:
    """Frame k of a smooth sequence: the bend travels one period over the sequence (phase 2 pi k / n), the camera follows
    sequence_gt_pose, matches are redrawn per frame (problem id = 100000 (seq_id + 1) + k).  init_* = the previous
    frame's result (warm start); None starts from the rest shape / identity like frame 0 of the reference."""
    return make_frame(tmpl, n_matches, 100000 * (seq_id + 1) + k, phase=2.0 * np.pi * k / n_frames - 0.3 * (100000 * (seq_id + 1) + k),
                      init_xyz=init_xyz, init_Tcw=init_Tcw, gt_pose=sequence_gt_pose(k, n_frames))


def make_problem(config: str = "C2", problem_id: int = 0):
    rows, cols, m = CONFIGS[config]
    tmpl = make_grid_template(rows, cols)
    return tmpl, make_frame(tmpl, m, problem_id)


# ---------------------------------------------------------------------------------------------------------
# NRSfM normals: a rigid planar scene seen from several keyframes.  For a plane N.X = 1 (camera-1 frame) the
# inter-image warp in normalised coordinates is the homography (R + t N^T); its first and second derivatives
# are the DiffProp fields (Modules/Mapping/diffProp.h, SchwarpDatabase.cc:299-345) and the solution of the
# two bicubic polynomials is k = (N1, N2) / (N.[u,v,1])  (normal ~ [k1, k2, 1 - k1 u - k2 v]).
# ---------------------------------------------------------------------------------------------------------
def _homography_derivs(Hm, u, v):
    p = np.array([u, v, 1.0])
    h = Hm @ p
    ha, hb = Hm[:, 0], Hm[:, 1]
    eta = h[:2] / h[2]
    d1 = {}
    for name, hd in (("u", ha), ("v", hb)):
        d1[name] = (hd[:2] * h[2] - h[:2] * hd[2]) / h[2] ** 2
    d2 = {}
    for (na, hd_a), (nb, hd_b) in ((("u", ha), ("u", ha)), (("u", ha), ("v", hb)), (("v", hb), ("v", hb))):
        d2[na + nb] = -(hd_a[:2] * hd_b[2] + hd_b[:2] * hd_a[2]) / h[2] ** 2 + 2 * h[:2] * hd_a[2] * hd_b[2] / h[2] ** 3
    return eta, d1, d2


def make_normals_scene(n_points: int = 200, n_views: int = 4, seed: int = 7, nonref_frac: float = 0.3, min_views: int = 1):
    """Returns a dict with the flat arrays of dsh_normals_estimate plus the ground-truth (k1,k2) per point.
    Every point gets between min_views and n_views records (the reference puts no limit on them, NormalEstimator.cc:77-118)."""
    rng = np.random.default_rng(seed)
    recs, is_ref, first_n, has_first_n, rec_ptr = [], [], [], [], [0]
    x0, has_x0, ref_uv, truth = [], [], [], []
    for p in range(n_points):
        # plane in the reference camera frame: N.X = 1, roughly fronto-parallel at depth ~1
        n = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), 1.0])
        N = n / rng.uniform(0.8, 1.5)
        u, v = rng.uniform(-0.4, 0.4, size=2)
        k_true = N[:2] / (N @ np.array([u, v, 1.0]))
        nv = int(rng.integers(min_views, n_views + 1))
        for _ in range(nv):
            w = rng.normal(size=3)
            w *= rng.uniform(0.03, 0.15) / np.linalg.norm(w)
            R = _rodrigues(w)
            t = rng.uniform(-0.15, 0.15, size=3)
            Hm = R + np.outer(t, N)
            eta, d1, d2 = _homography_derivs(Hm, u, v)
            a, b = d1["u"]      # d eta_u/du, d eta_v/du
            c, d = d1["v"]
            det = a * d - c * b
            rec = [u, v, eta[0], eta[1], a, b, c, d, d / det, -c / det, -b / det, a / det,
                   d2["uu"][0], d2["uu"][1], d2["uv"][0], d2["uv"][1], d2["vv"][0], d2["vv"][1]]
            # J21 fields follow SchwarpDatabase.cc:322-329: J21a = J12d/det, J21b = -J12c/det, J21c = -J12b/det, J21d = J12a/det
            recs.append(rec)
            ref = rng.uniform() >= nonref_frac
            is_ref.append(1 if ref else 0)
            hf = (not ref) and rng.uniform() < 0.7
            has_first_n.append(1 if hf else 0)
            first_n.append(list(k_true + rng.normal(scale=1e-3, size=2)) if hf else [0.0, 0.0])
        rec_ptr.append(len(recs))
        hx = rng.uniform() < 0.5
        has_x0.append(1 if hx else 0)
        x0.append(list(k_true + rng.normal(scale=0.05, size=2)) if hx else [0.0, 0.0])
        ref_uv.append([u, v])
        truth.append(k_true)
    return dict(rec_ptr=np.asarray(rec_ptr, np.int32), recs=np.asarray(recs, np.float32).reshape(-1, 18), rec_is_ref=np.asarray(is_ref, np.uint8),
                rec_first_normal=np.asarray(first_n, np.float32).reshape(-1, 2), rec_has_first_normal=np.asarray(has_first_n, np.uint8),
                x0=np.asarray(x0, np.float32), has_x0=np.asarray(has_x0, np.uint8), ref_uv=np.asarray(ref_uv, np.float32),
                k_true=np.asarray(truth))


# ---------------------------------------------------------------------------------------------------------
# Schwarp fit: matches between two keyframes related by a smooth warp (a homography of a plane plus a gentle
# non-rigid ripple), normalised image coordinates, 13 x 15 control grid as in Thirdparty/BBS/bbs_MAC.h.
# ---------------------------------------------------------------------------------------------------------
def _coloc_dense(umin, umax, nu, vmin, vmax, nv, u, v):
    """Dense colocation matrix of the uniform bicubic B-spline (numpy, for building test inputs only)."""
    def parts(x, xmin, xmax, npts):
        t = (x - xmin) * (npts - 3) / (xmax - xmin)
        i = np.minimum(np.floor(t).astype(int), npts - 4)
        t = t - i
        B = np.stack([(1 - t) ** 3, 3 * t**3 - 6 * t**2 + 4, -3 * t**3 + 3 * t**2 + 3 * t + 1, t**3], 1) / 6.0
        return i, B
    Iu, Bu = parts(np.asarray(u, float), umin, umax, nu)
    Iv, Bv = parts(np.asarray(v, float), vmin, vmax, nv)
    A = np.zeros((len(Iu), nu * nv))
    for a in range(4):
        for b in range(4):
            np.add.at(A, (np.arange(len(Iu)), (Iu + a) * nv + Iv + b), Bu[:, a] * Bv[:, b])
    return A


def make_warp_problem(n_matches: int = 400, seed: int = 3, nu: int = 13, nv: int = 15, noise: float = 5e-4, outliers: float = 0.0, kp1=None, motion=None):
    """One keyframe pair: matches kp1 -> kp2 under a plane-induced homography plus a ripple.  kp1 (given key points of the first keyframe,
    e.g. a subset of one pool shared by several pairs) and motion (rotation vector, translation of the second camera) are optional."""
    rng = np.random.default_rng(seed)
    if kp1 is None:
        kp1 = np.stack([rng.uniform(-0.55, 0.55, n_matches), rng.uniform(-0.42, 0.42, n_matches)], 1)
    else:
        kp1 = np.asarray(kp1, np.float64).reshape(-1, 2)
        n_matches = kp1.shape[0]
    N = np.array([0.15, -0.1, 1.0]) / 1.1
    rv, tv = motion if motion is not None else (np.array([0.02, -0.05, 0.03]), np.array([0.06, -0.04, 0.02]))
    Hm = _rodrigues(np.asarray(rv, np.float64)) + np.outer(np.asarray(tv, np.float64), N)
    p = np.c_[kp1, np.ones(n_matches)] @ Hm.T
    kp2 = p[:, :2] / p[:, 2:3]
    kp2[:, 0] += 0.01 * np.sin(3.0 * kp1[:, 1])
    kp2 += rng.normal(scale=noise, size=kp2.shape)
    if outliers > 0:
        bad = rng.uniform(size=n_matches) < outliers
        kp2[bad] += rng.uniform(-0.2, 0.2, size=(int(bad.sum()), 2))
    # domain: bounding box of the key points +- 0.10 (DefKeyFrame.cc:116-131)
    umin, umax = float(kp1[:, 0].min() - 0.10), float(kp1[:, 0].max() + 0.10)
    vmin, vmax = float(kp1[:, 1].min() - 0.10), float(kp1[:, 1].max() + 0.10)
    octave = rng.integers(0, 6, n_matches)
    invsig = np.sqrt((1.2 ** (-2.0 * octave)).astype(np.float32)).astype(np.float32)
    # start: what Warp::initialize hands over (Schwarp.cc:99-160) -- a regularised linear least-squares fit of the control
    # points to the matches; here the regulariser pulls towards the identity map (Greville abscissae).
    C = _coloc_dense(umin, umax, nu, vmin, vmax, nv, kp1[:, 0], kp1[:, 1])
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    su, sv = (umax - umin) / (nu - 3), (vmax - vmin) / (nv - 3)
    ident = np.stack([(umin + su * (iu - 1)).ravel(), (vmin + sv * (iv - 1)).ravel()], 1)
    mu = 1e-2
    cp = np.linalg.solve(C.T @ C + mu * np.eye(nu * nv), C.T @ kp2 + mu * ident)
    x0 = np.concatenate([cp[:, 0], cp[:, 1]])
    return dict(bbs=(umin, umax, nu, vmin, vmax, nv, 2), kp1=kp1.astype(np.float32), kp2=kp2.astype(np.float32), invsig=invsig, x0=x0,
                fx=520.0, fy=515.0)


# ---------------------------------------------------------------------------------------------------------
# Shape from Normals: a smooth depth surface z(u, v) over the normalised image plane; key points with the true
# surface normals (plus noise) as NRSfM would hand them over.  X(u, v) = z (u, v, 1): normal ~ X_u x X_v.
def make_sfn_scene(n_points: int = 600, seed: int = 4, noise: float = 0.01, with_normal_frac: float = 0.8):
    rng = np.random.default_rng(seed)
    u = rng.uniform(-0.55, 0.55, n_points)
    v = rng.uniform(-0.42, 0.42, n_points)

    def z(u, v):
        return 1.0 + 0.15 * u - 0.1 * v + 0.08 * np.sin(2.5 * u) * np.cos(2.0 * v)

    def zu(u, v):
        return 0.15 + 0.08 * 2.5 * np.cos(2.5 * u) * np.cos(2.0 * v)

    def zv(u, v):
        return -0.1 - 0.08 * 2.0 * np.sin(2.5 * u) * np.sin(2.0 * v)

    d = z(u, v)
    Xu = np.stack([zu(u, v) * u + d, zu(u, v) * v, zu(u, v)], 1)
    Xv = np.stack([zv(u, v) * u, zv(u, v) * v + d, zv(u, v)], 1)
    nrm = np.cross(Xu, Xv)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm += rng.normal(scale=noise, size=nrm.shape)
    has = rng.uniform(size=n_points) < with_normal_frac
    umin, umax = float(u.min() - 0.10), float(u.max() + 0.10)
    vmin, vmax = float(v.min() - 0.10), float(v.max() + 0.10)
    return dict(bbs=(umin, umax, 13, vmin, vmax, 15, 1), u_all=u, v_all=v, depth_true=d, u=u[has], v=v[has], normals=nrm[has].astype(np.float32),
                mean_depth=float(d.mean()))


# ---------------------------------------------------------------------------------------------------------
# Warp-guided match search: keyframe 1 key points (normalised) with ORB-like 256-bit descriptors, a warp, and keyframe 2
# key points scattered around the predictions (pixels) with noisy copies of the descriptors, distractors, exact duplicates
# (distance ties), points that already carry a map point and points outside the image / the grid.
def make_match_scene(n_query: int = 600, n_extra: int = 900, seed: int = 2):
    rng = np.random.default_rng(seed)
    pr = make_warp_problem(400, seed + 1)
    bbs = pr["bbs"][:6] + (2,)
    x = pr["x0"]
    cam2 = np.array([520.0, 515.0, 322.5, 241.25], np.float32)
    bounds2 = np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    kp1 = np.stack([rng.uniform(bbs[0] + 0.02, bbs[1] - 0.02, n_query), rng.uniform(bbs[3] + 0.02, bbs[4] - 0.02, n_query)], 1).astype(np.float32)
    # predictions with numpy (double) only to PLACE key points of keyframe 2; the tests compare device and oracle, not this
    C = _coloc_dense(bbs[0], bbs[1], bbs[2], bbs[3], bbs[4], bbs[5], kp1[:, 0].astype(float), kp1[:, 1].astype(float))
    N = bbs[2] * bbs[5]
    pred = np.stack([C @ x[:N], C @ x[N:]], 1)
    pix = pred * cam2[:2] + cam2[2:]
    desc1 = rng.integers(0, 256, (n_query, 32), dtype=np.uint8)
    kp2, desc2 = [], []
    for q in range(n_query):
        k = int(rng.integers(0, 4))                       # 0..3 candidates near the prediction
        for _ in range(k):
            kp2.append(pix[q] + rng.uniform(-2.6, 2.6, 2))
            d = desc1[q].copy()
            flips = rng.integers(0, 256, int(rng.integers(0, 70)))
            for f in flips:
                d[f // 8] ^= np.uint8(1 << (f % 8))
            desc2.append(d)
        if k and rng.uniform() < 0.25:                    # exact duplicate of the last candidate close by: a distance tie
            kp2.append(kp2[-1] + rng.uniform(-0.4, 0.4, 2))
            desc2.append(desc2[-1].copy())
    kp2 += list(np.stack([rng.uniform(-20, 660, n_extra), rng.uniform(-20, 500, n_extra)], 1))
    desc2 += list(rng.integers(0, 256, (n_extra, 32), dtype=np.uint8))
    kp2 = np.asarray(kp2, np.float32)
    desc2 = np.asarray(desc2, np.uint8)
    perm = rng.permutation(kp2.shape[0])
    kp2, desc2 = kp2[perm], desc2[perm]
    has_mp2 = (rng.uniform(size=kp2.shape[0]) < 0.2).astype(np.uint8)
    return dict(bbs=bbs, x=x, kp1=kp1, desc1=desc1, cam2=cam2, bounds2=bounds2, kp2=kp2, desc2=desc2, has_mp2=has_mp2)


def make_register_scene(n: int = 600, seed: int = 5, noise: float = 2e-3, outliers: float = 0.05, scale: float = 1.35):
    """Surface registration (SurfaceRegistration::registerSurfaces): a keyframe surface (up to scale, in world coordinates) and
    the map points it has to be aligned with, the keyframe's inverse pose and a stream of uniform numbers for scaleMinMedian."""
    rng = np.random.default_rng(seed)
    surf = np.stack([rng.uniform(-0.4, 0.4, n), rng.uniform(-0.3, 0.3, n), 1.0 + 0.15 * rng.standard_normal(n)], 1)
    R = _rodrigues(np.array([0.02, -0.035, 0.015]))
    t = np.array([0.03, -0.02, 0.05])
    mp = scale * (surf @ R.T) + t + noise * rng.standard_normal((n, 3))
    bad = rng.random(n) < outliers
    mp[bad] += 0.2 * rng.standard_normal((int(bad.sum()), 3))
    Rwc = _rodrigues(np.array([0.1, 0.05, -0.07]))
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, :3] = Rwc.astype(np.float32)
    Twc[:3, 3] = np.array([0.2, -0.1, 0.3], np.float32)
    u = rng.random(n + n * n)
    return dict(surface=surf.astype(np.float32), map=mp.astype(np.float32), Twc=Twc, u=u, scale=scale, R=R, t=t, outlier=bad)


# ---------------------------------------------------------------------------------------------------------
# One coherent mapping scene for the chained NRSfM loop (DefLocalMapping::NRSfM, DefLocalMapping.cc:160-234 and
# SchwarpDatabase::add): a smooth surface seen from a reference keyframe and `n_kf - 1` later keyframes.  Key points of
# the reference keyframe carry ORB-like descriptors; the first `n_tracked` of them are already matched in the later
# keyframes (tracked map points), the others are what the warp-guided search has to find.
# ---------------------------------------------------------------------------------------------------------
def make_mapping_scene(n_points: int = 520, n_tracked: int = 380, n_kf: int = 3, seed: int = 11, scale_true: float = 1.3):
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 520.0, 515.0, 322.5, 241.25
    u = rng.uniform(-0.5, 0.5, n_points)
    v = rng.uniform(-0.38, 0.38, n_points)
    depth = 1.0 + 0.12 * u - 0.08 * v + 0.05 * np.sin(2.0 * u) * np.cos(1.5 * v)
    X = np.stack([u * depth, v * depth, depth], 1)                      # reference camera frame
    desc0 = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    octave = rng.integers(0, 6, n_points)
    invsig = np.sqrt((1.2 ** (-2.0 * octave)).astype(np.float32)).astype(np.float32)
    kfs = []
    for j in range(1, n_kf):
        R = _rodrigues(np.array([0.03 * j, -0.05 * j, 0.02]) * (1.0 + 0.1 * rng.standard_normal()))
        t = np.array([0.05 * j, -0.03 * j, 0.02 * j])
        Xc = X @ R.T + t
        kp = Xc[:, :2] / Xc[:, 2:3] + rng.normal(scale=3e-4, size=(n_points, 2))
        pix = kp * np.array([fx, fy]) + np.array([cx, cy])
        desc = desc0.copy()
        for i in range(n_points):                                      # noisy copies: a few flipped bits
            for f in rng.integers(0, 256, int(rng.integers(0, 24))):
                desc[i, f // 8] ^= np.uint8(1 << (f % 8))
        n_extra = 300                                                   # distractors without a counterpart
        pix_all = np.vstack([pix, np.stack([rng.uniform(0, 640, n_extra), rng.uniform(0, 480, n_extra)], 1)])
        desc_all = np.vstack([desc, rng.integers(0, 256, (n_extra, 32), dtype=np.uint8)])
        perm = rng.permutation(pix_all.shape[0])
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.shape[0])
        has_mp = np.zeros(pix_all.shape[0], np.uint8)
        has_mp[inv[:n_tracked]] = 1                                     # tracked map points already own their key point
        kfs.append(dict(R=R, t=t, kp_norm=kp.astype(np.float32), pix=pix_all[perm].astype(np.float32), desc=desc_all[perm], index_of_point=inv[:n_points],
                        has_mp=has_mp))
    umin, umax = float(u.min() - 0.10), float(u.max() + 0.10)
    vmin, vmax = float(v.min() - 0.10), float(v.max() + 0.10)
    # the keyframe's pose in the map and the map points the surface is registered against (another scale, small noise)
    Rwc = _rodrigues(np.array([0.06, 0.03, -0.04]))
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, :3] = Rwc.astype(np.float32)
    Twc[:3, 3] = np.array([0.1, -0.05, 0.2], np.float32)
    Xw = scale_true * (X @ Rwc.T) + Twc[:3, 3].astype(np.float64)
    map_pts = (Xw + 1e-3 * rng.standard_normal(Xw.shape)).astype(np.float32)
    return dict(bbs2=(umin, umax, 13, vmin, vmax, 15, 2), bbs1=(umin, umax, 13, vmin, vmax, 15, 1), kp0=np.stack([u, v], 1).astype(np.float32), desc0=desc0,
                invsig=invsig, depth=depth, X=X, kfs=kfs, n_tracked=n_tracked, cam=np.array([fx, fy, cx, cy], np.float32),
                bounds=np.array([0.0, 640.0, 0.0, 480.0], np.float32), Twc=Twc, map_pts=map_pts, scale_true=scale_true,
                u_stream=rng.random(n_points + n_points * n_points))


# ---------------------------------------------------------------------------------------------------------
# SEQMAP (BASELINE.json configs[2] substitute, whole: deformable tracking AND NRSfM mapping in one sequence).
# The reference interleaves the two (DefTracking.cc:109-115,175; DefLocalMapping.cc:138-153,172-234): every
# 10th frame becomes a keyframe, the mapping side fits a Schwarzian warp from the anchor keyframe to it,
# re-estimates the normals of the anchor's map points, integrates them to a surface, registers the surface to
# the map and hands tracking a NEW template; the next frame is solved against it with RegTemp = 0.
# Scene: the anchor keyframe sees a smooth surface; it deforms slowly (a travelling bump along the normal
# direction) while the camera moves; all keyframes observe the same physical points.
# ---------------------------------------------------------------------------------------------------------
SEQMAP = dict(n_frames=41, kf_every=10, n_points=520, n_tracked=380, seed=17, scale_true=1.3, mesh=(12, 14))


def make_interleaved_sequence(n_frames: int = 41, kf_every: int = 10, n_points: int = 520, n_tracked: int = 380, seed: int = 17, scale_true: float = 1.3,
                              mesh=(12, 14)):
    """Everything the tracking + mapping loop consumes, frame by frame (ground truth included for plausibility checks only)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 520.0, 515.0, 322.5, 241.25
    u = rng.uniform(-0.5, 0.5, n_points)
    v = rng.uniform(-0.38, 0.38, n_points)

    def depth_of(uu, vv):
        return 1.0 + 0.12 * uu - 0.08 * vv + 0.05 * np.sin(2.0 * uu) * np.cos(1.5 * vv)

    depth = depth_of(u, v)
    X0 = np.stack([u * depth, v * depth, depth], 1)                     # anchor keyframe's camera frame, unit scale
    desc0 = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    octave = rng.integers(0, 6, n_points)
    invsig = np.sqrt((1.2 ** (-2.0 * octave)).astype(np.float32)).astype(np.float32)
    Rwc = _rodrigues(np.array([0.06, 0.03, -0.04]))
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, :3] = Rwc.astype(np.float32)
    Twc[:3, 3] = np.array([0.1, -0.05, 0.2], np.float32)
    twc = Twc[:3, 3].astype(np.float64)

    def to_world(Xa):                                                   # anchor frame (unit scale) -> world (map scale)
        return scale_true * (Xa @ Rwc.T) + twc

    def deformed(Xa, uu, vv, k):                                        # the surface at frame k: a slow travelling bump along the optical axis
        a = 0.012 * k / max(n_frames - 1, 1)
        out = Xa.copy()
        out[:, 2] += a * np.sin(3.0 * uu + 0.08 * k) * np.cos(2.0 * vv)
        return out

    Rcw0 = Rwc.T
    tcw0 = -Rcw0 @ twc
    frames = []
    for k in range(n_frames):
        dR = _rodrigues(np.array([0.004 * k, -0.006 * k, 0.002 * k]))
        Rk = dR @ Rcw0
        tk = dR @ tcw0 + scale_true * np.array([0.006 * k, -0.004 * k, 0.003 * k])
        T = np.eye(4)
        T[:3, :3] = Rk
        T[:3, 3] = tk
        Xw = to_world(deformed(X0, u, v, k))
        Xc = Xw @ Rk.T + tk
        frames.append(dict(Tcw_gt=T, Xw=Xw, Xc=Xc))
    kfs = {}
    # key -1: the keyframe the map was bootstrapped with before the sequence starts (a second view of the undeformed surface with a
    # decent baseline: one warp alone leaves the normals of the anchor poorly constrained); keys k = kf_every, 2 kf_every, ...: the sequence's own
    Rb = _rodrigues(np.array([0.06, -0.10, 0.02]))
    Xc_boot = X0 @ Rb.T + np.array([0.10, -0.06, 0.04])
    for k in [-1] + list(range(kf_every, n_frames, kf_every)):          # the keyframes' own measurements of the anchor's points
        Xc = Xc_boot if k < 0 else frames[k]["Xc"]
        kp = Xc[:, :2] / Xc[:, 2:3] + rng.normal(scale=3e-4, size=(n_points, 2))
        pix = kp * np.array([fx, fy]) + np.array([cx, cy])
        desc = desc0.copy()
        for i in range(n_points):
            for f in rng.integers(0, 256, int(rng.integers(0, 24))):
                desc[i, f // 8] ^= np.uint8(1 << (f % 8))
        n_extra = 300
        pix_all = np.vstack([pix, np.stack([rng.uniform(0, 640, n_extra), rng.uniform(0, 480, n_extra)], 1)])
        desc_all = np.vstack([desc, rng.integers(0, 256, (n_extra, 32), dtype=np.uint8)])
        perm = rng.permutation(pix_all.shape[0])
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.shape[0])
        has_mp = np.zeros(pix_all.shape[0], np.uint8)
        has_mp[inv[:n_tracked]] = 1
        kfs[k] = dict(kp_norm=kp.astype(np.float32), pix=pix_all[perm].astype(np.float32), desc=desc_all[perm], index_of_point=inv[:n_points], has_mp=has_mp,
                      u_stream=rng.random(n_points + n_points * n_points))
    umin, umax = float(u.min() - 0.10), float(u.max() + 0.10)
    vmin, vmax = float(v.min() - 0.10), float(v.max() + 0.10)
    # the first template (what MonocularInitialization + the first mapping pass leave behind): the anchor's true surface on a regular grid
    rows, cols = mesh
    gu, gv = np.meshgrid(np.linspace(u.min(), u.max(), cols), np.linspace(v.min(), v.max(), rows))
    gd = depth_of(gu.ravel(), gv.ravel())
    nodes0 = to_world(np.stack([gu.ravel() * gd, gv.ravel() * gd, gd], 1))
    noise = rng.normal(scale=0.4, size=(n_frames, n_points, 2))        # pixel noise of the tracked observations
    return dict(bbs2=(umin, umax, 13, vmin, vmax, 15, 2), bbs1=(umin, umax, 13, vmin, vmax, 15, 1), kp0=np.stack([u, v], 1).astype(np.float32), desc0=desc0,
                invsig=invsig, depth=depth, X0=X0, frames=frames, kfs=kfs, n_tracked=n_tracked, cam=np.array([fx, fy, cx, cy], np.float32),
                bounds=np.array([0.0, 640.0, 0.0, 480.0], np.float32), Twc=Twc, scale_true=scale_true, nodes0=nodes0, mesh=mesh, grid_uv=(gu, gv),
                facets=regular_triangulation(rows, cols), noise=noise, n_frames=n_frames, kf_every=kf_every)
