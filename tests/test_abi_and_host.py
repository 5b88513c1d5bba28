"""CPU tests: the C-ABI library loads, exports every symbol the header declares, and the host-side
logic (template constants, embedding, problem packing) matches the oracle.  No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from defslam_amd import _lib
    L = _lib.load()
    header = open(os.path.join(ROOT, "include", "defslam_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dsh_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no prototypes found in the header"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/defslam_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def _dynamic_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if " T " in line)


def test_product_library_exports_exactly_the_public_header_and_the_lab_library_adds_the_debug_header():
    """The product ABI carries no lab equipment: libdefslam_hip.so exports the prototypes of include/defslam_hip.h and nothing
    else (no timers, no test hooks, no internal launchers), and does not read the environment; the measurement / debugging
    entry points of include/defslam_hip_debug.h exist only in libdefslam_hip_lab.so (the same sources with -DDSH_LAB)."""
    import subprocess
    from defslam_amd import _lib
    header = open(os.path.join(ROOT, "include", "defslam_hip.h")).read()
    debug_header = open(os.path.join(ROOT, "include", "defslam_hip_debug.h")).read()
    declared = sorted(set(re.findall(r"\b(dsh_[a-z0-9_]+)\s*\(", header)))
    lab_declared = sorted(set(re.findall(r"\b(dsh_lab_[a-z0-9_]+)\s*\(", debug_header)))
    assert lab_declared and sorted(_lib.LAB_SYMBOLS) == lab_declared
    assert not re.search(r"timed|phase_ms|debug_system|dsh_lab", header)
    assert _dynamic_symbols(_lib.LIB_PATH) == declared
    assert _dynamic_symbols(_lib.LAB_LIB_PATH) == sorted(declared + lab_declared)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined, "the product library must not read environment switches"
    L = _lib.load_lab()
    for name in lab_declared:
        assert hasattr(L, name)


def test_template_set_refuses_malformed_index_arrays(host_ctx):
    """dsh_template_set turns the caller's CSR / edge arrays into indices of the packer: out-of-range or non-monotone
    input is DSH_ERR_ARG, not a heap overrun (nothing aborts across the ABI)."""
    from defslam_amd import sft, synth
    tmpl = synth.make_grid_template(6, 7)
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    t = host_ctx.template_get()

    def attempt(**over):
        a = dict(t)
        a.update(over)
        host_ctx.template_set(tmpl.xyz0, a["boundary"], a["nbr_ptr"], a["nbr_idx"], a["nbr_w"], a["k0"], a["edge_nodes"], a["edge_L0"], a["median_L"])

    attempt()   # the constants the library derived itself are accepted
    bad_ptr = t["nbr_ptr"].copy(); bad_ptr[3] = bad_ptr[2] - 1
    bad_ptr0 = t["nbr_ptr"].copy(); bad_ptr0[0] = 1
    bad_col = t["nbr_idx"].copy(); bad_col[5] = tmpl.n
    neg_col = t["nbr_idx"].copy(); neg_col[0] = -1
    bad_edge = t["edge_nodes"].copy(); bad_edge[4, 1] = 10_000
    bad_len = t["edge_L0"].copy(); bad_len[2] = 0.0
    for over in (dict(nbr_ptr=bad_ptr), dict(nbr_ptr=bad_ptr0), dict(nbr_idx=bad_col), dict(nbr_idx=neg_col), dict(edge_nodes=bad_edge),
                 dict(edge_L0=bad_len), dict(median_L=0.0), dict(median_L=float("nan"))):
        with pytest.raises(sft.DshError, match="dsh_template_set"):
            attempt(**over)
    attempt()


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """The boundary is a C ABI: include/defslam_hip.h compiles as C99 and as C++11, and a C program linked against the
    shared library creates and destroys a host-only context (what a cgo / JNI / N-API stub would do first)."""
    import subprocess
    from defslam_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "client.c"
    src.write_text('#include "include/defslam_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { dsh_ctx* c = 0; if (dsh_create(&c, -1) != DSH_OK) return 1;\n'
                   '  if (dsh_synchronize(c) == DSH_OK) return 2;   /* host-only context: GPU entry points refuse */\n'
                   '  printf("%s\\n", dsh_last_error(c)); return dsh_destroy(c); }\n')
    lib = _lib.LIB_PATH
    exe = tmp_path / "client"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", root, str(src), "-o", str(exe), lib,
                    f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", root, str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gpu_entry_points_fail_loudly_without_a_device(host_ctx):
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem("smoke")
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    f = sft.frame_from_synth(fr)
    host_ctx.batch_upload([f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    with pytest.raises(sft.DshError, match="no GPU|host-only"):
        host_ctx.batch_run()


def test_mapping_and_registration_entry_points_fail_loudly_without_a_device(host_ctx):
    """No CPU fallback anywhere: the one-shot calls refuse a host-only context too (and bad arguments are a status, not a crash)."""
    from defslam_amd import nrsfm, register, sft, synth
    sc = synth.make_register_scene(40, seed=1)
    for call in (lambda: register.scaleMinMedian(host_ctx, sc["surface"], sc["map"], sc["u"]),
                 lambda: register.OptimizeHorn(host_ctx, sc["surface"], sc["map"], [0, 0, 0, 1, 0, 0, 0, 1.0], 0.01),
                 lambda: register.registerSurfaces(host_ctx, sc["surface"], sc["map"], sc["u"], sc["Twc"], 0.05),
                 lambda: nrsfm.bbs_eval(host_ctx, nrsfm.Bbs(0, 1, 5, 0, 1, 5, 1), np.zeros((25, 1)), np.array([0.5]), np.array([0.5]))):
        with pytest.raises(sft.DshError, match="no GPU|host-only|NO_DEVICE|status"):
            call()
    tmpl = synth.make_grid_template(5, 5)
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    with pytest.raises(sft.DshError, match="host-only"):
        host_ctx.template_embed_device(np.zeros((3, 3), np.float32))
    # fewer than 15 pairs is the reference's early `return false`, answered without touching a device
    few = register.registerSurfaces(host_ctx, sc["surface"][:10], sc["map"][:10], sc["u"], sc["Twc"], 0.05)
    assert not few["registered"]
    with pytest.raises(sft.DshError):
        register.scaleMinMedian(host_ctx, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), sc["u"])


def test_bad_arguments_return_status_not_abort(host_ctx):
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem("smoke")
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    f = sft.frame_from_synth(fr)
    f.obs_nodes = f.obs_nodes.copy()
    f.obs_nodes[0, 0] = 10_000
    with pytest.raises(sft.DshError, match="out of range"):
        host_ctx.batch_upload([f])
    bad_facets = tmpl.facets.copy()
    bad_facets[0, 0] = -1
    with pytest.raises(sft.DshError):
        host_ctx.template_build(tmpl.xyz0, bad_facets)


@pytest.mark.parametrize("shape", [(10, 10), (25, 20), (7, 13)])
def test_template_constants_match_oracle(host_ctx, oracle_mod, shape):
    from defslam_amd import synth
    tmpl = synth.make_grid_template(*shape)
    tc = oracle_mod.template_build(tmpl.xyz0, tmpl.facets)
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    tg = host_ctx.template_get()
    # integer / index work: bit exact
    for k in ["boundary", "nbr_ptr", "nbr_idx", "edge_nodes"]:
        np.testing.assert_array_equal(tg[k], getattr(tc, k))
    # floating point: same formulas in the same order -> identical doubles
    np.testing.assert_array_equal(tg["edge_L0"], tc.edge_L0)
    np.testing.assert_array_equal(tg["nbr_w"], tc.nbr_w)
    np.testing.assert_array_equal(tg["k0"], tc.k0)
    assert tg["median_L"] == tc.median_L
    rows, cols = shape
    assert tc.E == (rows - 1) * cols + (cols - 1) * rows + (rows - 1) * (cols - 1)
    assert int(tc.boundary.sum()) == 2 * (rows + cols) - 4


def test_irregular_mesh_constants(host_ctx, oracle_mod):
    """A fan + strip mesh: varying degrees, shuffled facet orientation."""
    rng = np.random.default_rng(5)
    xy = rng.uniform(-1, 1, size=(40, 2))
    from scipy.spatial import Delaunay
    tri = Delaunay(xy)
    xyz = np.c_[xy, 1 + 0.1 * rng.normal(size=40)]
    fac = tri.simplices.astype(np.int32)
    tc = oracle_mod.template_build(xyz, fac)
    host_ctx.template_build(xyz, fac)
    tg = host_ctx.template_get()
    for k in ["boundary", "nbr_ptr", "nbr_idx", "edge_nodes"]:
        np.testing.assert_array_equal(tg[k], getattr(tc, k))
    np.testing.assert_array_equal(tg["nbr_w"], tc.nbr_w)
    np.testing.assert_array_equal(tg["k0"], tc.k0)


def test_embedding_matches_oracle_bit_exact(host_ctx, oracle_mod):
    from defslam_amd import synth
    tmpl = synth.make_grid_template(10, 10)
    tc = oracle_mod.template_build(tmpl.xyz0, tmpl.facets)
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    rng = np.random.default_rng(7)
    F = tmpl.facets.shape[0]
    fac = rng.integers(0, F, size=500)
    bary = rng.dirichlet((1, 1, 1), size=500)
    pts = (bary[:, :, None] * tmpl.xyz0[tmpl.facets[fac]]).sum(1)
    pts[::7] += rng.normal(scale=0.01, size=pts[::7].shape)       # off-surface points
    pts[::50] += 5.0                                               # far away: not embedded
    pts = pts.astype(np.float32)
    fid, nodes, b = host_ctx.template_embed(pts)
    L = oracle_mod.lib()
    ofid = np.zeros(500, np.int32)
    ob = np.zeros((500, 3), np.float32)
    xyz0 = np.ascontiguousarray(tmpl.xyz0)
    L.tmpl_oracle_embed(tc.n, xyz0.ctypes.data_as(C.POINTER(C.c_double)), F, tc.facets.ctypes.data_as(C.POINTER(C.c_int32)), 500,
                        pts.ctypes.data_as(C.POINTER(C.c_float)), ofid.ctypes.data_as(C.POINTER(C.c_int32)), ob.ctypes.data_as(C.POINTER(C.c_float)))
    np.testing.assert_array_equal(fid, ofid)            # bit-exact triangle indexing
    np.testing.assert_array_equal(b, ob)                # float32 barycentrics, same operation order
    ok = fid >= 0
    np.testing.assert_array_equal(nodes[ok], tc.facets[fid[ok]])
    assert (fid[::50] == -1).all() and ok.sum() > 400
    # reconstruct: sum b_k v_k is within the reference's 0.1 squared-distance gate
    rec = (b[ok][:, :, None] * tmpl.xyz0[nodes[ok]]).sum(1)
    assert ((rec - pts[ok]) ** 2).sum(1).max() <= 0.1


def test_packer_counts_and_algorithmic_bytes(host_ctx, oracle_mod):
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem("C2")
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    host_ctx.batch_upload([sft.frame_from_synth(fr)], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    nbytes, counts = host_ctx.problem_info(0)
    tc = oracle_mod.template_build(tmpl.xyz0, tmpl.facets)
    r = oracle_mod.sft_solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz,
                             synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, max_iters=0)
    D, nopt, nview, ncurv, nstr, _ = r.dims
    assert list(counts[:6]) == [1000, nopt, ncurv, nstr, nview, D]
    assert 0 < counts[6] <= 128 and counts[7] == 8     # half-bandwidth of the node block (tile mode), latency launch shape for one problem
    M, n, Cc, E, V = 1000, 500, int(ncurv), int(nstr), int(nview)
    expect = (60 * M + 24 * n + 88 + 92 * Cc + 16 * E + 28 * V) + 8 * (30 * M + 21 * Cc + 6 * E + 9 * V) + 8 * (2 * M + Cc + E + 3 * V) + 8 * M
    assert nbytes == expect
    assert 1.1e6 < nbytes < 1.25e6    # ~1.17 MB per assembly pass at C2


def test_graph_cache_survives_more_active_sets_than_it_holds(host_ctx):
    """The structure of the normal equations is cached per active set (64 entries).  A sequence whose view changes every frame, and a
    batch with more distinct active sets than the cache holds, must keep packing correctly: the eviction may only drop graphs the
    upload in progress does not use (regression: it used to clear the problems already packed by the same upload)."""
    from defslam_amd import sft, synth
    tmpl = synth.make_grid_template(14, 14)
    host_ctx.template_build(tmpl.xyz0, tmpl.facets)
    base = synth.make_frame(tmpl, 900, 3)

    def windowed(r0, c0):
        keep = [c + 14 * r for r in range(r0, r0 + 6) for c in range(c0, c0 + 6)]
        sel = np.all(np.isin(base.obs_nodes, keep), axis=1)
        f = sft.frame_from_synth(base)
        f.obs_nodes, f.obs_bary, f.obs_uv, f.obs_invsig2 = base.obs_nodes[sel], base.obs_bary[sel], base.obs_uv[sel], base.obs_invsig2[sel]
        return f

    views = [(r, c) for r in range(9) for c in range(9)]          # 81 different 6x6 windows = 81 active sets
    ref = {}
    for v in views:                                                # one frame at a time: the 65th evicts
        host_ctx.batch_upload([windowed(*v)], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 10)
        ref[v] = host_ctx.problem_info(0)
        assert ref[v][1][0] > 0 and 0 < ref[v][1][1] <= 64         # some observations, at most the 8x8 nodes of window + ring
    # all of them in ONE upload (more active sets than the cache holds), in a different order
    order = views[::-1]
    host_ctx.batch_upload([windowed(*v) for v in order], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 10)
    for b, v in enumerate(order):
        nb, counts = host_ctx.problem_info(b)
        assert nb == ref[v][0]
        np.testing.assert_array_equal(counts[:7], ref[v][1][:7])
        assert counts[8] == ref[v][1][8]
