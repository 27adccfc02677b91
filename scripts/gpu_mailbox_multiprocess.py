"""The node-local mailbox ACROSS PROCESSES: `world` separate processes -- one HIP context each, all on device 0 because a box of
the pool has one GPU; RCCL is not involved -- shard one Vecchia evaluation, every process's finisher workgroup writes its 3 / 7 sums into the shared-memory
segment (shm_open + hipHostRegister in every process), every host polls all slots.  Checks: all ranks return identical bits; the job's sums equal the
ranks' shard sums added in rank order, bit for bit; they agree with the unsharded evaluation to 1e-12.  Reports the time per evaluation.
(The in-process rank groups of tests/test_multirank_gpu.py exercise the same protocol inside one process; this is the form bench.py --gpus N uses.)
    python scripts/gpu_mailbox_multiprocess.py [world ...]        default: 2 4"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, D, M = 400000, 2, 30
PARS = [(10.0, 9.0 + 0.25 * k) for k in range(6)]


def data():
    rng = np.random.default_rng(7)
    return rng.uniform(size=(N, D)), rng.standard_normal(N)


def worker(rank, world, conns, barrier, out_q):
    try:
        import gpboost_amd
        from gpboost_amd import shim, parallel
        gpboost_amd.set_device(0)
        co, y = data()
        st = shim.VecchiaState(co, M)
        st.find_neighbors()
        st.set_y(y)
        i0, i1 = parallel.shard_range(N, rank, world)
        st.set_shard(i0, i1)
        if rank == 0:
            name = shim.mailbox_create(world)
            for c in conns:
                c.send(name)
        else:
            name = conns.recv()
        st.mailbox_attach(name, rank, world)
        barrier.wait()
        vals = [np.asarray(st.nll_terms_allreduce(0, v, a)).copy() for v, a in PARS]
        grads = [np.asarray(st.grad_terms_allreduce(0, v, a)).copy() for v, a in PARS[:2]]
        barrier.wait()
        t0 = time.perf_counter()
        for k in range(200):
            st.nll_terms_allreduce(0, PARS[k % len(PARS)][0], PARS[k % len(PARS)][1])
        ms = (time.perf_counter() - t0) / 200 * 1e3
        barrier.wait()
        st.mailbox_detach()
        st.set_shard(i0, i1)
        local = [np.asarray(st.nll_terms(0, v, a)).copy() for v, a in PARS]
        out_q.put((rank, "ok", vals, grads, local, ms, st.mailbox_info() if False else None))
    except Exception as e:   # noqa: BLE001
        out_q.put((rank, "error: %s: %s" % (type(e).__name__, e), None, None, None, 0.0, None))
        try:
            barrier.abort()
        except Exception:   # noqa: BLE001
            pass


def run(world):
    ctx = mp.get_context("spawn")
    barrier = ctx.Barrier(world)
    out_q = ctx.Queue()
    pipes = [ctx.Pipe() for _ in range(world - 1)]
    procs = []
    for r in range(world):
        conns = [p[0] for p in pipes] if r == 0 else pipes[r - 1][1]
        procs.append(ctx.Process(target=worker, args=(r, world, conns, barrier, out_q)))
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = out_q.get(timeout=600)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    bad = [v[1] for v in res.values() if v[1] != "ok"]
    if bad:
        raise RuntimeError("; ".join(bad))
    for k in range(len(PARS)):
        for r in range(1, world):
            assert np.array_equal(res[r][2][k], res[0][2][k]), ("ranks differ", k, r)
        want = np.zeros(3)
        for r in range(world):
            want = want + res[r][4][k][:3]
        assert np.array_equal(res[0][2][k][:3], want), ("not the rank-ordered sum", k, res[0][2][k][:3], want)
    for k in range(2):
        for r in range(1, world):
            assert np.array_equal(res[r][3][k], res[0][3][k]), ("gradient sums differ between ranks", k, r)
    return res


if __name__ == "__main__":
    worlds = [int(a) for a in sys.argv[1:]] or [2, 4]
    import gpboost_amd
    from gpboost_amd import shim
    gpboost_amd.set_device(0)
    co, y = data()
    st = shim.VecchiaState(co, M)
    st.find_neighbors()
    st.set_y(y)
    full = [np.asarray(st.nll_terms(0, v, a)).copy() for v, a in PARS]
    t0 = time.perf_counter()
    for k in range(200):
        st.nll_terms(0, PARS[k % len(PARS)][0], PARS[k % len(PARS)][1])
    ms1 = (time.perf_counter() - t0) / 200 * 1e3
    del st
    print("one process, unsharded: %.4f ms per evaluation (n = %d, m = %d)" % (ms1, N, M), flush=True)
    for w in worlds:
        res = run(w)
        for k in range(len(PARS)):
            np.testing.assert_allclose(res[0][2][k][:2], full[k][:2], rtol=1e-12)
        print("%d processes on one device through the mailbox: identical bits on every rank, = the rank-ordered sum of the shard sums, = the unsharded "
              "evaluation to 1e-12; %.4f ms per evaluation (max over ranks; the ranks share ONE device here, so their kernels run one after the other)"
              % (w, max(v[5] for v in res.values())), flush=True)
    print("MAILBOX ACROSS PROCESSES: OK")
