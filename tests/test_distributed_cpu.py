"""CPU, world_size 2 over gloo: the multi-GPU composition of the path -- contiguous shards of the Vecchia
ordering per rank, one all-reduce of the partial likelihood/gradient terms -- with the oracle
standing in for the per-shard kernel.  Checks gpboost_amd.parallel's sharding and reduction logic."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gpboost_amd import parallel
    from oracle import orc
    from tests import cases
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coords, y = cases.synthetic(700, 2, seed=5)
    perm, co, nn = orc.vecchia_setup(coords, 12, "random", 1)
    pt = orc.transform_cov_pars(0, np.array([0.1, 1.0, 0.1]))
    A, D, Ag, Dg, bad = orc.vecchia_factor(co, nn, 0, pt[1], pt[2], grad=True)
    yv = y[perm]
    i0, i1 = parallel.shard_range(len(yv), rank, world)
    # per-shard terms exactly as the kernel accumulates them (vecchia_kernels.h GPB_P_*)
    u = yv - np.einsum("ij,ij->i", A, np.where(nn >= 0, yv[np.maximum(nn, 0)], 0.))
    up = u / D
    uk = [-np.einsum("ij,ij->i", Ag[p], np.where(nn >= 0, yv[np.maximum(nn, 0)], 0.)) for p in range(2)]
    sl = slice(i0, i1)
    t7 = np.array([np.sum(u[sl] ** 2 / D[sl]), np.sum(np.log(D[sl])), 0.,
                   np.sum(uk[0][sl] * up[sl] - 0.5 * up[sl] ** 2 * Dg[0][sl]), np.sum(0.5 * Dg[0][sl] / D[sl]),
                   np.sum(uk[1][sl] * up[sl] - 0.5 * up[sl] ** 2 * Dg[1][sl]), np.sum(0.5 * Dg[1][sl] / D[sl])])
    tot = parallel.allreduce_terms(torch.from_numpy(t7.copy())).numpy()
    nll = parallel.nll_from_terms(len(yv), tot[0], tot[1], pt[0])
    grad = parallel.grad_from_terms(len(yv), tot, pt[0])
    out, g = orc.vecchia_nll_grad(co, nn, 0, pt, yv)
    q.put((rank, i0, i1, float(nll), float(out[2]), grad.tolist(), g.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_nll_and_gradient_match_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 700    # contiguous cover
    for r in res:
        assert abs(r[3] - r[4]) <= 1e-10 * abs(r[4])
        np.testing.assert_allclose(r[5], r[6], rtol=1e-9, atol=1e-9)


def _worker_other(rank, world, port, q):
    """The other sharded pieces of SURVEY.md 8e with the oracle standing in for the kernels: data-parallel histograms (rows per
    rank, sum-all-reduce), y_aux (row shards of B, sum-all-reduce of n doubles) and the neighbour search (blocks of positions in
    coordinate-sum order per rank, rows of other queries below -1, max-all-reduce)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gpboost_amd import parallel
    from oracle import orc
    from tests import cases
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # histograms
    rng = np.random.default_rng(3)
    n, F = 5000, 5
    nb = rng.integers(2, 257, size=F); bo = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    bins = np.stack([rng.integers(0, nb[f], size=n) for f in range(F)]).astype(np.uint8)
    g = rng.standard_normal(n)
    leaf = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int32)
    r0, r1 = parallel.shard_range(n, rank, world)
    mine = leaf[(leaf >= r0) & (leaf < r1)]
    hg, hc, hh = orc.hist_build(bins, bo, mine, g, None)
    tg = torch.from_numpy(hg.copy()); tc = torch.from_numpy(hc.astype(np.int64))
    dist.all_reduce(tg); dist.all_reduce(tc)
    fg, fc, fh = orc.hist_build(bins, bo, leaf, g, None)
    ok_hist = bool(np.array_equal(tc.numpy(), fc.astype(np.int64)) and np.allclose(tg.numpy(), fg, atol=1e-10))
    # y_aux and neighbour table
    coords, y = cases.synthetic(900, 2, seed=8)
    perm, co, nn = orc.vecchia_setup(coords, 10, "random", 1)
    A, D, bad = orc.vecchia_factor(co, nn, 0, 10.0, 10.0)
    yv = y[perm]
    i0, i1 = parallel.shard_range(len(yv), rank, world)
    u = yv - np.einsum("ij,ij->i", A, np.where(nn >= 0, yv[np.maximum(nn, 0)], 0.))
    v = np.zeros_like(u); v[i0:i1] = u[i0:i1] / D[i0:i1]
    w = v.copy()
    for i in range(i0, i1):
        for j in range(nn.shape[1]):
            if nn[i, j] >= 0:
                w[nn[i, j]] -= A[i, j] * v[i]
    tw = torch.from_numpy(w); dist.all_reduce(tw)
    ok_yaux = bool(np.allclose(tw.numpy(), orc.vecchia_yaux(A, D, nn, yv), atol=1e-10))
    order = orc.sort_indices(co.sum(axis=1))                       # coordinate-sum order
    p0, p1 = len(yv) * rank // world, len(yv) * (rank + 1) // world
    part = np.full(nn.shape, np.iinfo(np.int32).min, dtype=np.int32)
    rows = np.concatenate([order[p0:p1], np.arange(min(len(yv), nn.shape[1] + 1))])     # my block + the head rows
    part[rows] = nn[rows]
    tp = torch.from_numpy(part); dist.all_reduce(tp, op=dist.ReduceOp.MAX)
    ok_nn = bool(np.array_equal(tp.numpy(), nn))
    q.put((rank, ok_hist, ok_yaux, ok_nn))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_histogram_yaux_and_neighbor_table_compose():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_other, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3], r


def test_shard_range_covers_everything():
    from gpboost_amd import parallel
    for n in (1, 7, 16, 1000, 10 ** 6 + 3):
        for w in (1, 2, 3, 8):
            edges = [parallel.shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for a, b in zip(edges[:-1], edges[1:]):
                assert a[1] == b[0]
            sizes = [e[1] - e[0] for e in edges]
            assert max(sizes) - min(sizes) <= 16 * w


def _worker_fit(rank, world, port, q):
    """Sharded parameter estimation: every rank runs the product's host optimiser (GPB_HIP_OptimizeGaussianWithCallback); its
    evaluation callback returns the all-reduced (gloo) shard sums, the oracle standing in for the per-shard kernel."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    import torch.distributed as dist
    from gpboost_amd import parallel
    from gpboost_amd.libpath import find_lib_path
    from oracle import orc
    from tests import cases, optim_harness as oh
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name = "r_lbfgs_default"
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    g = np.load(os.path.join(ROOT, "tests", "golden", "optim_ref.npz"))
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    yv = y[perm]
    i0, i1 = parallel.shard_range(len(yv), rank, world)
    sl = slice(i0, i1)
    ynn = np.where(nn >= 0, yv[np.maximum(nn, 0)], 0.)

    def terms(ctx, ratio, a, with_grad, t7):
        A, D, Ag, Dg, bad = orc.vecchia_factor(co, nn, 0, ratio, a, grad=True)
        u = yv - np.einsum("ij,ij->i", A, ynn)
        up = u / D
        uk = [-np.einsum("ij,ij->i", Ag[p], ynn) for p in range(2)]
        loc = np.array([np.sum(u[sl] ** 2 / D[sl]), np.sum(np.log(D[sl])), 0.,
                        np.sum(uk[0][sl] * up[sl] - 0.5 * up[sl] ** 2 * Dg[0][sl]), np.sum(0.5 * Dg[0][sl] / D[sl]),
                        np.sum(uk[1][sl] * up[sl] - 0.5 * up[sl] ** 2 * Dg[1][sl]), np.sum(0.5 * Dg[1][sl] / D[sl])])
        tot = parallel.allreduce_terms(torch.from_numpy(loc)).numpy()
        for k in range(7):
            t7[k] = tot[k]
        return 0
    lib = C.CDLL(find_lib_path())
    th0 = orc.transform_cov_pars(0, g[name + "_init_cov_pars"])
    th, nit, nll, ne = oh.optimize(lib, len(yv), th0, oh.TERMS_FN(terms))
    q.put((rank, [th[0], th[1] * th[0], 1.0 / th[2]], nit, nll, g[name + "_cov_pars"].tolist(), int(g[name + "_num_it"]), float(g[name + "_negll"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_fit_follows_the_reference_trajectory():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fit, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] and res[0][3] == res[1][3]      # identical decisions on both ranks
    for r in res:
        assert r[2] == r[5]
        np.testing.assert_allclose(r[1], r[4], rtol=2e-6)
        assert abs(r[3] - r[6]) <= 1e-9 * abs(r[6])
