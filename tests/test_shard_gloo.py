"""CPU test of the N>1 path: world_size-2 gloo processes shard independent problems and gather results to rank 0.
The per-problem 'solve' here is a deterministic stand-in (the HIP solve needs a GPU); what is tested is the partition,
the ragged gather and the result order that bench.py / a multi-GPU host rely on."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_solve(ids):
    # a "result vector" whose length depends on the problem (ragged) and whose content identifies it
    return [np.arange(3 + (i % 4), dtype=np.float64) * 0.5 + 100.0 * i for i in ids]


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    from defslam_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = shard.solve_sharded(list(range(n_items)), _fake_solve, dist)
        if rank == 0:
            q.put([o.tolist() for o in out])
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_items", [7, 2, 1])
def test_two_rank_sharded_solve_gathers_in_id_order(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = [v.tolist() for v in _fake_solve(list(range(n_items)))]
    assert got == expect


def test_shard_range_partitions_everything_once():
    from defslam_amd import shard
    for n in [0, 1, 7, 8, 9, 256, 1000]:
        for world in [1, 2, 3, 8]:
            seen = []
            for r in range(world):
                rr = shard.shard_range(n, r, world)
                seen += list(rr)
                assert len(rr) in (n // world, n // world + 1)
            assert seen == list(range(n))


# ---- the shared-camera exchange protocol between two real processes --------------------------------------------------------
# Every rank owns one patch of the template (its nodes and observations); the camera is shared.  What crosses the process
# boundary is what dsh_sft_shared_solve all-reduces over RCCL: the rank's 6x6 Schur complement of the camera and its right-hand
# side.  Here the per-rank normal equations come from the oracle (no GPU in this container) and the all-reduce is gloo; the
# result must be the solution of the JOINT system the oracle builds for the union of the patches.
def _camera_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from defslam_amd import synth
    from test_shared_camera_gpu import _split_template
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tmpl = synth.make_grid_template(8, 14)
        fr = synth.make_frame(tmpl, 500, 6)
        rng = np.random.default_rng(3)
        fr.xyz = fr.xyz + rng.normal(scale=0.002, size=fr.xyz.shape)
        facets, patches = _split_template(tmpl, [7])
        on_patch = np.isin(np.sort(tmpl.facets[fr.obs_facet], axis=1).view([("", np.int32)] * 3).ravel(),
                           np.sort(facets, axis=1).view([("", np.int32)] * 3).ravel())
        for k in ["obs_facet", "obs_nodes", "obs_bary", "obs_uv", "obs_invsig2"]:
            setattr(fr, k, getattr(fr, k)[on_patch])
        regs = np.array([synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP])
        tcj = oracle.template_build(tmpl.xyz0, facets)
        # this rank's patch; the regulariser weights divide by the JOINT counts and the joint median edge length: scale the
        # patch's parameters so that its own normalisation gives the joint weights
        ids, lf = patches[rank]
        local = -np.ones(tmpl.n, np.int64)
        local[ids] = np.arange(ids.size)
        sel = np.all(local[fr.obs_nodes] >= 0, axis=1)
        tcp = oracle.template_build(tmpl.xyz0[ids], lf)
        counts = torch.tensor([float(ids.size), float(tcp.E)], dtype=torch.float64)   # every node of a patch is optimised here (all viewed or 1-ring)
        tot = counts.clone()
        dist.all_reduce(tot)
        scale = np.array([counts[0] / tot[0], counts[1] / tot[1], (tcp.median_L / tcj.median_L) ** 2])
        Hp, bp, chip = oracle.sft_system(tcp, fr.Tcw, fr.K, fr.n_frame, local[fr.obs_nodes[sel]].astype(np.int32), fr.obs_bary[sel], fr.obs_uv[sel],
                                         fr.obs_invsig2[sel], fr.xyz[ids], *(regs * scale))
        assert Hp.shape[0] == 6 + 3 * ids.size
        lam = 1e-3 * np.abs(np.diag(Hp)).max()
        lam_t = torch.tensor([lam], dtype=torch.float64)
        dist.all_reduce(lam_t, op=dist.ReduceOp.MAX)
        lam = float(lam_t.item())
        Hnn = Hp[6:, 6:] + lam * np.eye(3 * ids.size)
        Hcn = Hp[:6, 6:]
        S = Hp[:6, :6] + (lam * np.eye(6) if rank == 0 else 0.0) - Hcn @ np.linalg.solve(Hnn, Hcn.T)
        rhs = bp[:6] - Hcn @ np.linalg.solve(Hnn, bp[6:])
        buf = torch.from_numpy(np.concatenate([S[np.tril_indices(6)], rhs, [chip]]))   # 21 + 6 + 1 = what the ranks exchange
        dist.all_reduce(buf)
        St = np.zeros((6, 6))
        St[np.tril_indices(6)] = buf[:21].numpy()
        St = St + np.tril(St, -1).T
        xc = np.linalg.solve(St, buf[21:27].numpy())
        xn = np.linalg.solve(Hnn, bp[6:] - Hcn.T @ xc)
        # the joint system of the union of the patches (every rank builds it: the test's reference)
        Hj, bj, chij = oracle.sft_system(tcj, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs)
        xj = np.linalg.solve(Hj + lam * np.eye(Hj.shape[0]), bj)
        rows = 6 + 3 * np.repeat(ids, 3) + np.tile(np.arange(3), ids.size)
        q.put((rank, float(np.abs(xc - xj[:6]).max() / np.abs(xj[:6]).max()), float(np.abs(xn - xj[rows]).max() / np.abs(xj).max()),
               float(abs(buf[27].item() - chij) / chij)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shared_camera_exchange_protocol_between_two_processes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_camera_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1]
    for _, err_cam, err_nodes, err_chi in got:
        assert err_cam < 1e-9 and err_nodes < 1e-9 and err_chi < 1e-12


# ---- the connected-mesh protocol between two real processes ---------------------------------------------------------------------
# dsh_sft_connected_solve: ONE connected template, the band ordering cut at a separator of one bandwidth; rank g eliminates part g, the
# ranks all-reduce their Schur contributions to the separator + camera system, solve it, back-substitute their part, and all-reduce the
# pieces of the update.  Here the normal equations of the connected mesh come from the oracle (no GPU in this container), the two
# all-reduces are gloo, and the result must be the solution of the undivided system.
def _connected_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import oracle
    from defslam_amd import sft, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tmpl = synth.make_grid_template(14, 9)            # connected 14 x 9 grid: nothing is dropped at the cut
        fr = synth.make_frame(tmpl, 500, 11)
        fr.xyz = fr.xyz + np.random.default_rng(5).normal(scale=0.002, size=fr.xyz.shape)
        tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
        regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
        H, b, chi = oracle.sft_system(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs)
        D = H.shape[0]
        Dn = D - 6
        Hn = H[6:, 6:]
        rr, cc = np.nonzero(Hn)
        kd = int(np.abs(rr - cc).max())                   # scalar half-bandwidth of the node block (all nodes active: natural order)
        c0, s, n1p, pad = sft.two_sided_cut(Dn, kd)
        assert s >= kd
        lam = 1e-3 * np.abs(np.diag(H)).max()
        Hd = H + lam * np.eye(D)
        cam = np.arange(6)
        sep = 6 + np.arange(c0, c0 + s)
        mine = 6 + (np.arange(0, c0) if rank == 0 else np.arange(c0 + s, Dn))
        other = 6 + (np.arange(c0 + s, Dn) if rank == 0 else np.arange(0, c0))
        assert np.abs(H[np.ix_(mine, other)]).max() == 0.0          # the separator decouples the two parts
        red = np.concatenate([sep, cam])                  # the reduced unknowns: separator, then camera
        Hgg = Hd[np.ix_(mine, mine)]
        Hrg = Hd[np.ix_(red, mine)]
        Sg = -Hrg @ np.linalg.solve(Hgg, Hrg.T)
        rg = -Hrg @ np.linalg.solve(Hgg, b[mine])
        if rank == 0:                                     # rank 0 carries the separator block, the camera corner (with their damping) and their right-hand side
            Sg = Sg + Hd[np.ix_(red, red)]
            rg = rg + b[red]
        buf = torch.from_numpy(np.concatenate([Sg[np.tril_indices(red.size)], rg]))
        dist.all_reduce(buf)                              # first collective: the Schur contributions
        nt = red.size * (red.size + 1) // 2
        S = np.zeros((red.size, red.size))
        S[np.tril_indices(red.size)] = buf[:nt].numpy()
        S = S + np.tril(S, -1).T
        xr = np.linalg.solve(S, buf[nt:].numpy())
        xg = np.linalg.solve(Hgg, b[mine] - Hrg.T @ xr)   # back substitution of the rank's own part
        x = np.zeros(D)
        x[mine] = xg
        if rank == 0:
            x[red] = xr
        xt = torch.from_numpy(x)
        dist.all_reduce(xt)                               # second collective: the pieces of the update
        xj = np.linalg.solve(Hd, b)
        q.put((rank, float(np.abs(xt.numpy() - xj).max() / np.abs(xj).max()), int(kd), int(s), int(c0)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_connected_mesh_exchange_protocol_between_two_processes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_connected_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1]
    for _, err, kd, s, c0 in got:
        assert err < 1e-9 and s >= kd and c0 > 0
