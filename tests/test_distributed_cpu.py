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
