"""GPU (one MI355X): the SHARDED code paths with several ranks.

RCCL wants one device per rank, and a test box has one device.  The library's collectives therefore have a second transport, an
in-process group whose ranks are threads (include/gpb_hip.h: gpb_hip_local_group_create): same host code, same kernels, same buffers,
only the all-reduce itself is "publish, barrier, reduce every rank's buffer in rank order, barrier".  With it the data-parallel
histogram / tree learner (DataParallelTreeLearner's scheme, data_parallel_tree_learner.cpp:155-173, :240-260) and the sharded Vecchia
evaluation run here with 2, 3 and 4 ranks on real kernels.

What is asserted is stronger than "close": the job's histograms, counts and trees are BIT-IDENTICAL to the one-handle result on the
same rows for every number of ranks and every way of dealing the rows to them (one fixed-point scale agreed by an all-reduce(max);
integer totals on the wire, converted once)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n, F, seed):
    rng = np.random.default_rng(seed)
    nb = rng.integers(2, 257, size=F)
    nb[0] = 256
    bo = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    bins = np.stack([rng.integers(0, nb[f], size=n) for f in range(F)]).astype(np.uint8)
    grad = rng.standard_normal(n) * np.exp(rng.uniform(-6, 6, size=n))       # 5 decades of magnitudes: the ranks' OWN maxima differ widely
    return bins, bo, grad, rng.uniform(0.5, 2.0, size=n), rng


def _deal(n, world, how, rng):
    if how == "blocks":
        cuts = np.linspace(0, n, world + 1).astype(int)
        return [np.arange(cuts[r], cuts[r + 1]) for r in range(world)]
    if how == "uneven":
        cuts = np.concatenate([[0], np.sort(rng.choice(np.arange(1, n), size=world - 1, replace=False)), [n]])
        return [np.arange(cuts[r], cuts[r + 1]) for r in range(world)]
    owner = rng.integers(0, world, size=n)                                    # "random": rows dealt at random (ascending within a rank)
    return [np.flatnonzero(owner == r) for r in range(world)]


@pytest.mark.parametrize("world,how", [(2, "blocks"), (3, "uneven"), (4, "random")])
@pytest.mark.parametrize("with_hess", [False, True])
def test_sharded_histograms_equal_the_one_handle_histogram_bit_for_bit(lib_built, world, how, with_hess):
    from gpboost_amd import shim
    n, F = 300000, 37
    bins, bo, grad, hess, rng = _case(n, F, seed=11 + world)
    hs = hess if with_hess else None
    one = shim.HistBuilder(bins, bo); one.set_gradients(grad, hs)
    leaf_mask = rng.uniform(size=n) < 0.4
    ref_all = one.build(None, const_hess=0.7)
    ref_leaf = one.build(np.flatnonzero(leaf_mask).astype(np.int32), const_hess=0.7)
    one.close()
    parts = _deal(n, world, how, rng)
    grp = shim.LocalGroup(world)

    def rank(r):
        rows = parts[r]
        hb = shim.HistBuilder(np.ascontiguousarray(bins[:, rows]), bo)
        hb.comm_init_local(grp, r)
        hb.set_gradients(grad[rows], None if hs is None else hs[rows])       # collective: the scale is agreed here
        a = hb.build_allreduce(None, const_hess=0.7)
        b = hb.build_allreduce(np.flatnonzero(leaf_mask[rows]).astype(np.int32), const_hess=0.7)
        hb.close()
        return a, b
    res = grp.run(rank)
    for a, b in res:
        assert np.array_equal(a[1], ref_all[1]) and np.array_equal(b[1], ref_leaf[1]), "counts"
        assert np.array_equal(a[0], ref_all[0]), "whole-data histogram differs from the one-handle histogram"
        assert np.array_equal(b[0], ref_leaf[0]), "leaf histogram differs from the one-handle histogram"
    grp.close()


@pytest.mark.parametrize("name,hi,world,how", [("plain_l31", 0, 2, "blocks"), ("plain_l31", 1, 3, "random"), ("nan_l20", 0, 4, "uneven"),
                                               ("zero_missing_l12", 1, 2, "random"), ("plain_all_reg", 0, 3, "blocks"),
                                               # round 5: categorical features (their sets of bins travel in the exchange), more ranks than features (empty blocks)
                                               ("cat_l15", 0, 3, "random"), ("cat_defaults", 1, 2, "blocks"), ("efb_l15", 0, 4, "uneven"), ("plain_l15_reg", 1, 8, "random")])
@pytest.mark.parametrize("exchange", ["feature_blocks", "allreduce"])
def test_sharded_tree_equals_the_one_rank_tree_bit_for_bit(lib_built, name, hi, world, how, exchange):
    """gpb_hip_hist_grow_tree in its data-parallel form on the reference's tree fixtures' data: W ranks (rows dealt W ways) return, on
    every rank, the tree a ONE-rank group returns -- every field bit-identical, leaf values included -- and that tree has the structure,
    thresholds and counts of the reference's own SerialTreeLearner tree (tests/golden/tree_ref.npz).  Both exchanges of the data-parallel learner:
    'feature_blocks' (round 5, the default: reduce-scatter of the integer totals by feature block, every rank searches its own features, the ranks' best
    splits are exchanged -- data_parallel_tree_learner.cpp:155-173, :244) and 'allreduce' (every rank all-reduces every histogram and searches everything)."""
    from gpboost_amd import shim
    from tests import cases
    r5 = name in cases.TREE_CASES_R5
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree_ref_r5.npz" if r5 else "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    n = X.shape[0]
    k = "%s_hess%d_" % (name, hi)
    hs = hess if hi else None
    bins, gnb, mfb, meta3 = g[k + "bins"], g[k + "group_num_bin"], g[k + "most_freq_bin"], g[k + "meta3"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    rng = np.random.default_rng(3)

    def grow(W, parts):
        grp = shim.LocalGroup(W)

        def rank(r):
            rows = parts[r]
            hb = shim.HistBuilder(np.ascontiguousarray(bins[:, rows]), bo)
            hb.pool_resize(L + 1)
            hb.set_fix_info(g[k + "view_offset"], g[k + "num_bin"], mfb)
            hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
            if r5 and g[k + "layout"][:, 3].any():
                hb.set_categorical(g[k + "layout"][:, 3], *cases.tree_cat_cfg(name))
            hb.comm_init_local(grp, r)
            hb.set_feature_block_exchange(exchange == "feature_blocks")
            hb.set_gradients(grad[rows], None if hs is None else hs[rows])
            if len(cfg) > 4:
                hb.set_regularisation(cfg[4], cfg[5], cfg[6])
            hb.set_max_depth(cases.tree_max_depth(name))
            t = hb.grow_tree(L, float("nan"), float("nan"), *cfg[:4])         # root sums: from the all-reduced integer totals
            hb.close()
            return t
        out = grp.run(rank)
        grp.close()
        return out
    t1 = grow(1, [np.arange(n)])[0]
    parts = _deal(n, world, how, rng)
    tw = grow(world, parts)
    keys = ("split_feature_inner", "threshold_in_bin", "default_left", "left_child", "right_child", "internal_count", "leaf_count", "split_gain",
            "leaf_value", "node_is_cat", "node_cat_bits")
    for r, t in enumerate(tw):
        assert t["num_leaves"] == t1["num_leaves"]
        for key in keys:
            assert np.array_equal(t[key], t1[key]), (key, r)
        assert np.array_equal(t["data_leaf_index"], t1["data_leaf_index"][parts[r]]), "row labels of rank %d" % r
    # ... and it is the reference's tree
    assert t1["num_leaves"] == int(g[k + "num_leaves"])
    for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count"):
        assert np.array_equal(t1[key], g[k + key]), key
    np.testing.assert_allclose(t1["leaf_value"], g[k + "leaf_value"], rtol=1e-9, atol=1e-12)


def test_sharded_vecchia_evaluation_neighbours_yaux_with_three_ranks(lib_built, orc):
    """SURVEY.md 8e rows 1-3 with 3 ranks on the device: neighbour-search parts merged by the max-all-reduce, likelihood + gradient terms
    of contiguous point shards summed by the all-reduce, y_aux contributions summed -- against the unsharded handle and the oracle."""
    from gpboost_amd import parallel, shim
    n, m, W = 30011, 30, 3
    rng = np.random.default_rng(4)
    coords = rng.uniform(size=(n, 2)); y = rng.standard_normal(n)
    var, a = 10.0, 1.0 / 0.1
    full = shim.VecchiaState(coords, m); full.find_neighbors(); full.set_y(y)
    nn = full.get_neighbors()
    t3 = full.nll_terms(0, var, a); t7 = full.grad_terms(0, var, a)
    full.factor(0, var, a); ya = full.yaux()
    full.close()
    grp = shim.LocalGroup(W)

    def rank(r):
        st = shim.VecchiaState(coords, m)
        st.comm_init_local(grp, r)
        assert st.comm_info() == (r, W)
        st.find_neighbors_part(r, W)
        st.neighbors_allreduce()
        nn_r = st.get_neighbors()
        st.set_y(y)
        st.set_shard(*parallel.shard_range(n, r, W))
        o3 = st.nll_terms_allreduce(0, var, a)
        o7 = st.grad_terms_allreduce(0, var, a)
        st.factor(0, var, a)
        ya_r = st.yaux_allreduce()
        st.close()
        return nn_r, o3, o7, ya_r
    for nn_r, o3, o7, ya_r in grp.run(rank):
        assert np.array_equal(nn_r, nn), "merged neighbour table"
        np.testing.assert_allclose(o3, t3, rtol=1e-12)
        np.testing.assert_allclose(o7, t7, rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(ya_r, ya, rtol=1e-10, atol=1e-12)
    grp.close()


@pytest.mark.parametrize("W", [2, 4])
def test_mailbox_sums_of_a_sharded_evaluation(lib_built, W):
    """SURVEY.md 8e row 1 through the node-local shared-memory MAILBOX (round 4): W ranks (threads, one device) attach to one
    segment, every rank's finisher workgroup stores its shard sums into its slot, every host polls all slots and adds them in rank order -- no
    collective launch.  Likelihood and gradient terms equal the unsharded handle's (1e-12), are bit-identical on all ranks, over many evaluations
    (the three slot generations are re-used), and alternate between 3 and 7 terms."""
    from gpboost_amd import parallel, shim
    n, m = 24013, 30
    rng = np.random.default_rng(7)
    coords = rng.uniform(size=(n, 2)); y = rng.standard_normal(n)
    pars = [(10.0 * (1 + 0.01 * k), 1.0 / (0.1 * (1 + 0.005 * k))) for k in range(9)]
    full = shim.VecchiaState(coords, m); full.find_neighbors(); full.set_y(y)
    nn = full.get_neighbors()
    ref3 = [full.nll_terms(0, v, a) for v, a in pars]
    ref7 = [full.grad_terms(0, v, a) for v, a in pars]
    full.close()
    name = shim.mailbox_create(W)
    grp = shim.LocalGroup(W)                 # (only to run the rank functions as threads; the sums do not use its all-reduce)

    def rank(r):
        st = shim.VecchiaState(coords, m)
        st.set_neighbors(nn)
        st.set_y(y)
        st.set_shard(*parallel.shard_range(n, r, W))
        st.mailbox_attach(name, r, W)
        assert st.mailbox_info() == (r, W) and st.comm_info() == (r, W)
        out = []
        for k, (v, a) in enumerate(pars):
            out.append(st.nll_terms_allreduce(0, v, a))
            if k % 2 == 0:
                out.append(st.grad_terms_allreduce(0, v, a))
        st.mailbox_detach()
        st.close()
        return out
    res = grp.run(rank)
    for r in range(1, W):
        for a_, b_ in zip(res[0], res[r]):
            assert np.array_equal(a_, b_), "every rank adds the slots in rank order: identical bits"
    q = 0
    for k in range(len(pars)):
        np.testing.assert_allclose(res[0][q], ref3[k], rtol=1e-12); q += 1
        if k % 2 == 0:
            np.testing.assert_allclose(res[0][q], ref7[k], rtol=1e-10, atol=1e-8); q += 1
    grp.close()


@pytest.mark.gpu
def test_mailbox_across_processes(lib_built):
    """The form bench.py --gpus N uses: separate PROCESSES (own HIP context each; all on device 0 here, no RCCL) shard one evaluation and exchange their sums
    through the shared-memory mailbox -- identical bits on every rank, equal to the rank-ordered sum of the shard sums and to the unsharded evaluation
    (scripts/gpu_mailbox_multiprocess.py; profiles/r04_n_mailbox_multiprocess.log)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_mailbox_multiprocess.py"), "2", "3"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MAILBOX ACROSS PROCESSES: OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 4])
def test_bench_starts_its_own_ranks_and_rehearses_on_one_device(lib_built, N):
    """`python bench.py --gpus N` with NO launcher around it (the way the driver started N = 1): the script starts its own N ranks through
    torch.distributed.run; on a box with fewer devices than ranks they rehearse on device 0 (gloo bootstrap, mailbox for the shard sums, RCCL down) and
    walk the code of a real N-device launch after the bootstrap.  The line must say n_gpus = N, name the rehearsal, report N mailbox ranks, shards that
    cover n, and the job's value must equal the one-process value of the same evaluation (the mailbox adds the shard sums in rank order: 1e-12)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    n = 200000

    def run(gpus):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "4", "--warmup", "1", "--n", str(n),
                            "--no-cpu-baseline", "--no-extras"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-3000:]
        return json.loads(lines[0])

    one = run(1)
    out = run(N)
    assert one["n_gpus"] == 1 and out["n_gpus"] == N
    cfg = out["config"]
    assert cfg["mailbox_ranks"] == N and cfg["per_rank"]["mailbox_ranks_seen"] == [N] * N
    assert sum(cfg["per_rank"]["shard_points"]) == n and len(cfg["per_rank"]["kernel_ms"]) == N
    import gpboost_amd
    if gpboost_amd.device_count() < N:
        assert cfg["rehearsal"] is True and cfg["rccl_ranks"] == 0 and "REHEARSAL" in out["metric"]
    else:
        assert cfg["rehearsal"] is False and cfg["rccl_ranks"] == N
    assert abs(cfg["last_negll"] - one["config"]["last_negll"]) <= 1e-12 * abs(one["config"]["last_negll"])
    assert out["roofline"]["kernel_ms"] <= out["ms_per_step"] and cfg["overhead_us"] >= 0.0


@pytest.mark.parametrize("name,hi", [("plain_l31", 0), ("cat_l15", 1)])
def test_every_collective_of_the_multi_gpu_path_runs_through_a_one_rank_rccl_communicator(lib_built, orc, name, hi):
    """Multi-GPU readiness without the hardware: the FIRST launch on a node with N > 1 devices must not meet an RCCL call that has never
    executed.  With a one-rank communicator from ncclCommInitRank (not the in-process LocalGroup transport of the tests above) this drives every collective the
    N > 1 path issues -- Vecchia: the neighbour table's max-all-reduce (ncclAllReduce, int32 max), the 3 / 7 shard sums (ncclAllReduce, fp64 sum), y_aux
    (ncclAllReduce over n doubles); trees: the fixed-point scale (max), a histogram's integer totals (ncclAllReduce, int64 sum), and BOTH exchanges of the
    data-parallel grower -- 'allreduce' and 'feature_blocks' (ncclReduceScatter of the packed integer totals + the all-reduce of the ranks' candidate records) --
    and every result must equal the communicator-free one bit for bit (the reference's scheme: data_parallel_tree_learner.cpp:131, 155-173, 244)."""
    from gpboost_amd import shim
    from tests import cases
    # ---- Vecchia side
    n, m = 20011, 20
    coords, y = cases.synthetic(n, 2, seed=5)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 2)
    st = shim.VecchiaState(co, m)
    st.comm_init(shim.comm_unique_id(), 0, 1)
    assert st.comm_info() == (0, 1)
    st.find_neighbors_part(0, 1)
    st.neighbors_allreduce()
    assert np.array_equal(st.get_neighbors(), nn)
    st.set_y(y[perm])
    assert np.array_equal(st.nll_terms_allreduce(0, 10.0, 10.0), st.nll_terms(0, 10.0, 10.0))
    assert np.array_equal(st.grad_terms_allreduce(0, 10.0, 10.0), st.grad_terms(0, 10.0, 10.0))
    st.factor(0, 10.0, 10.0)
    assert np.array_equal(st.yaux_allreduce(), st.yaux())
    st.close()
    # ---- tree side: the reference's tree fixtures' data
    r5 = name in cases.TREE_CASES_R5
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree_ref_r5.npz" if r5 else "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    k = "%s_hess%d_" % (name, hi)
    hs = hess if hi else None
    bins, gnb, mfb, meta3 = g[k + "bins"], g[k + "group_num_bin"], g[k + "most_freq_bin"], g[k + "meta3"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)

    def grow(mode):
        hb = shim.HistBuilder(np.ascontiguousarray(bins), bo)
        hb.pool_resize(L + 1)
        hb.set_fix_info(g[k + "view_offset"], g[k + "num_bin"], mfb)
        hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
        if r5 and g[k + "layout"][:, 3].any():
            hb.set_categorical(g[k + "layout"][:, 3], *cases.tree_cat_cfg(name))
        if mode is not None:
            hb.comm_init(shim.comm_unique_id(), 0, 1)
            hb.set_feature_block_exchange(mode == "feature_blocks")
        hb.set_gradients(grad, hs)
        if len(cfg) > 4:
            hb.set_regularisation(cfg[4], cfg[5], cfg[6])
        hb.set_max_depth(cases.tree_max_depth(name))
        h_all = hb.build_allreduce(None) if mode is not None else hb.build(None)
        sg = float("nan") if mode is not None else float(np.cumsum(grad)[-1])
        sh = float("nan") if mode is not None else (float(len(grad)) if hs is None else float(np.cumsum(hs)[-1]))
        t = hb.grow_tree(L, sg, sh, *cfg[:4])
        hb.close()
        return h_all, t
    h0, t0 = grow(None)
    for mode in ("allreduce", "feature_blocks"):
        h1, t1 = grow(mode)
        assert np.array_equal(h1[0], h0[0]) and np.array_equal(h1[1], h0[1]), mode
        assert t1["num_leaves"] == t0["num_leaves"]
        for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count", "node_is_cat", "node_cat_bits"):
            assert np.array_equal(t1[key], t0[key]), (mode, key)
        np.testing.assert_allclose(t1["leaf_value"], t0["leaf_value"], rtol=1e-10, atol=1e-13)
