"""Host integration shim (integration/): the reference's call sites compiled over the C ABI.

CPU: the shim and the writers compile (g++, stand-in types); Matches.txt / ErrorGTs files have the format scripts/Twiddle.py
reads back (parsed here the way Twiddle.py:38-131 does).  GPU: the compiled driver runs DefPoseOptimizationHIP on stand-in
Frame / DefMap objects and every in-place mutation of SURVEY 8b "Ownership" is checked against the Python mirror of the operator."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

INTEG = os.path.join(ROOT, "integration")


def _build():
    subprocess.run(["make", "-C", INTEG], check=True, capture_output=True)
    return os.path.join(INTEG, "build", "shim_test")


def test_shim_and_writers_compile_against_the_c_abi():
    exe = _build()
    assert os.path.exists(exe) and os.path.exists(os.path.join(INTEG, "build", "result_writers.o"))
    # the shim header only names the public ABI
    src = open(os.path.join(INTEG, "defslam_hip_shim.h")).read()
    assert "defslam_hip_debug.h" not in src and "dsh_lab" not in src


def test_writers_produce_what_twiddle_reads(tmp_path):
    import pandas as pd
    _build()
    prog = tmp_path / "w.cc"
    prog.write_text('#include "integration/result_writers.h"\n'
                    'int main(int, char** argv) { defslam_hip::MatchesWriter m(std::string(argv[1]) + "/Matches.txt");\n'
                    '  m.add_row(3, 410, 12, 640); m.add_row(12, 388, 40, 655); m.add_row(104, 0, 0, 700);\n'
                    '  std::vector<std::vector<float>> mono = {{0.1f, 0.2f, 1.0f}, {0.0f, -0.1f, 1.2f}, {0.3f, 0.1f, 0.9f}};\n'
                    '  std::vector<std::vector<float>> stereo = {{0.13f, 0.26f, 1.31f}, {0.0f, -0.13f, 1.55f}, {0.4f, 0.12f, 1.2f}};\n'
                    '  for (unsigned t : {3u, 12u}) { auto e = defslam_hip::surface_errors(mono, stereo, 1.3);\n'
                    '    if (!defslam_hip::save_results(e, defslam_hip::error_gts_name(argv[1], t))) return 1; }\n'
                    '  return m.ok() ? 0 : 1; }\n')
    exe = tmp_path / "w"
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", ROOT, str(prog), os.path.join(INTEG, "build", "result_writers.o"), "-o", str(exe)], check=True)
    subprocess.run([str(exe), str(tmp_path)], check=True)
    assert open(tmp_path / "Matches.txt").read() == "00003 410 12 640\n00012 388 40 655\n00104 0 0 700\n"
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ErrorGTs")) == ["ErrorGTs00003.txt", "ErrorGTs00012.txt"]
    # ---- read back exactly like scripts/Twiddle.py:38-131 (rms_per_sequence)
    df, names = None, []
    for file in sorted(os.listdir(tmp_path)):
        if file.startswith("ErrorGTs"):
            d = pd.read_csv(str(tmp_path / file), header=None).transpose()
            names.append(file.strip("ErrorGTs").strip(".txt"))
            df = d if df is None else pd.concat([df, d])
    df["frame"] = names
    df["frame"] = df["frame"].astype("int32")
    assert sorted(df["frame"].tolist()) == [3, 12]
    vals = df.iloc[:, 0:-1].to_numpy(dtype=float)
    mono = np.array([[0.1, 0.2, 1.0], [0.0, -0.1, 1.2], [0.3, 0.1, 0.9]], np.float32).astype(float)
    stereo = np.array([[0.13, 0.26, 1.31], [0.0, -0.13, 1.55], [0.4, 0.12, 1.2]], np.float32).astype(float)
    expect = np.linalg.norm(stereo - 1.3 * mono, axis=1)
    np.testing.assert_allclose(vals[0], expect, rtol=2e-6)              # six significant digits like Eigen's default stream format
    dm = pd.read_csv(str(tmp_path / "Matches.txt"), sep=" ", header=None, names=["frame", "inliers", "outliers", "possibleMatches"])
    assert dm["frame"].astype("int32").tolist() == [3, 12, 104]
    assert dm["inliers"].sum() / dm["possibleMatches"].sum() == pytest.approx((410 + 388) / (640 + 655 + 700))
    lines = open(tmp_path / "ErrorGTs00003.txt").read().split("\n")
    assert len(lines) == 3 and len({len(ln) for ln in lines}) == 1         # right-aligned to one width, no trailing newline


@pytest.mark.gpu
def test_def_pose_optimization_hip_mutates_frame_map_and_template_like_the_reference(gpu_ctx, tmp_path):
    from defslam_amd import sft, synth
    exe = _build()
    tmpl, fr = synth.make_problem("smoke", 3)
    rng = np.random.default_rng(5)
    M = fr.obs_nodes.shape[0]
    facets_sorted = np.sort(tmpl.facets, axis=1)
    # key points of the frame: the matches (kind 1) interleaved with key points the reference skips: no map point (0), bad map
    # point (2), map point without facet (3), flagged outlier (4)
    kinds = np.r_[np.ones(M, int), np.zeros(40, int), np.full(15, 2), np.full(10, 3), np.full(12, 4)]
    src = np.r_[np.arange(M), rng.integers(0, M, 77)]
    perm = rng.permutation(kinds.size)
    kinds, src = kinds[perm], src[perm]
    N = kinds.size
    levels = (1.2 ** (-2.0 * np.arange(8))).astype(np.float32)
    octave = np.round(-np.log(fr.obs_invsig2) / (2 * np.log(1.2))).astype(int)
    xyz_now = fr.xyz + rng.normal(scale=1e-3, size=fr.xyz.shape)
    with open(tmp_path / "in.txt", "w") as f:
        f.write(f"{tmpl.n} {facets_sorted.shape[0]}\n")
        for row in tmpl.xyz0:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        for row in xyz_now:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        for row in facets_sorted:
            f.write(" ".join(str(int(v)) for v in row) + "\n")
        f.write(" ".join(repr(float(v)) for v in fr.K) + "\n")
        f.write(" ".join(repr(float(v)) for v in fr.Tcw.ravel()) + "\n")
        f.write(f"{N} 37\n{levels.size}\n" + " ".join(repr(float(v)) for v in levels) + "\n")
        for i in range(N):
            m = src[i]
            f.write(f"{float(fr.obs_uv[m, 0])!r} {float(fr.obs_uv[m, 1])!r} {octave[m]} {kinds[i]} {fr.obs_facet[m]} "
                    + " ".join(repr(float(v)) for v in fr.obs_bary[m]) + "\n")
        f.write(f"{synth.REG_LAP!r} {synth.REG_INEX!r} {synth.REG_TEMP!r} 1\n640\n")
    r = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.txt"), str(tmp_path / "Matches.txt"), "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tok = open(tmp_path / "out.txt").read().split()
    it = iter(tok)
    inliers, rep, pose_sets = int(next(it)), float(next(it)), int(next(it))
    Tcw = np.array([float(next(it)) for _ in range(16)], np.float32).reshape(4, 4)
    node_rows = np.array([[float(next(it)) for _ in range(7)] for _ in range(tmpl.n)])
    outl = np.array([int(next(it)) for _ in range(N)], bool)
    n_mp = int((kinds != 0).sum())
    mp_rows = np.array([[float(next(it)) for _ in range(4)] for _ in range(n_mp)])
    # ---- the same call through the Python mirror: the key points the reference takes, in key point order
    taken = np.nonzero(kinds == 1)[0]
    f = sft.Frame(Tcw=fr.Tcw.copy(), K=fr.K.copy(), N=N, obs_nodes=facets_sorted[fr.obs_facet[src[taken]]].astype(np.int32), obs_bary=fr.obs_bary[src[taken]],
                  obs_uv=fr.obs_uv[src[taken]], obs_invsig2=levels[octave[src[taken]]].astype(np.float64), nodes_xyz=xyz_now.copy())
    gpu_ctx.template_build(tmpl.xyz0, facets_sorted)
    inl = sft.DefPoseOptimization(gpu_ctx, f, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    assert inliers == inl and pose_sets == 1                                 # return value; SetPose called once
    np.testing.assert_array_equal(Tcw, f.Tcw)                                # pFrame->mTcw
    assert np.float32(rep) == np.float32(f.repError)                         # pFrame->repError
    np.testing.assert_array_equal(node_rows[:, :3], f.nodes_xyz)            # Node::x,y,z of every node
    np.testing.assert_array_equal(node_rows[:, 3], np.arange(1, tmpl.n + 1)) # Node::indx = vertex id (setMeshNodes)
    assert (node_rows[:, 6] == 0).all()                                      # roles reset after the update (updateNodes)
    viewed = np.zeros(tmpl.n, bool)
    viewed[np.unique(f.obs_nodes)] = True
    np.testing.assert_array_equal(node_rows[:, 4].astype(bool), viewed)      # Node::viewed
    assert ((node_rows[:, 5] == 1) & viewed).sum() == 0 and (node_rows[:, 5] == 1).sum() > 0   # 1-ring of the viewed zone is LOCAL
    # pFrame->mvbOutlier: written for the key points that entered the graph, untouched for every other key point
    np.testing.assert_array_equal(outl[taken], f.mvbOutlier)
    assert outl[kinds == 4].all() and not outl[kinds == 0].any() and not outl[kinds == 2].any() and not outl[kinds == 3].any()
    # DefMapPoint::mWorldPos: RecalculatePosition of every map point with a facet (also bad ones and flagged ones), exactly once
    has_facet = kinds[kinds != 0] != 3
    assert (mp_rows[has_facet, 3] == 1).all() and (mp_rows[~has_facet, 3] == 0).all()
    mp_src = src[kinds != 0][has_facet]
    expect = (fr.obs_bary[mp_src][:, :, None] * f.nodes_xyz[facets_sorted[fr.obs_facet[mp_src]]]).sum(1).astype(np.float32)
    np.testing.assert_allclose(mp_rows[has_facet, :3], expect, rtol=0, atol=2e-7)
    # Matches.txt row of the frame (DefTracking.cc:299-328): inliers / outliers among key points with a good map point
    good = (kinds == 1) | (kinds == 3) | (kinds == 4)
    mI, mO = int((~outl[good]).sum()), int(outl[good].sum())
    assert open(tmp_path / "Matches.txt").read() == f"00037 {mI} {mO} 640\n"
