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
    "C2": (25, 20, 1000),   # 500-node template, 1000 matches (BASELINE.json configs[1])
    "C5": (50, 40, 4000),   # 2000-node template, 4000 matches (configs[4], per problem)
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
               init_Tcw: np.ndarray | None = None, phase: float = 0.0) -> SftFrame:
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


def make_problem(config: str = "C2", problem_id: int = 0):
    rows, cols, m = CONFIGS[config]
    tmpl = make_grid_template(rows, cols)
    return tmpl, make_frame(tmpl, m, problem_id)
