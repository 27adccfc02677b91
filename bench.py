#!/usr/bin/env python
"""bench.py -- neg-log-likelihood evaluations per second of the Gaussian Vecchia GP on MI355X.

Metric (BASELINE.json): neg-log-lik evals/sec, n = 1e6, Vecchia m = 30, fp64, on 1/2/4/8 GPUs; % of roofline.
A "step" is ONE evaluation of the whole job's likelihood at fresh covariance parameters: every rank runs the
fused point kernel on its contiguous shard of the Vecchia ordering (inputs resident in HBM), the <= 3 partial
sums are all-reduced over RCCL (N > 1), and the value is delivered to the host (as an optimiser would need it).
Model creation (ordering + device neighbour search) is set-up and reported separately, as in BASELINE.md.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (vecchia_point_kernel, MODE_NLL): its
average duration is measured live with HIP events on the stream it is launched on (gpb_hip_vecchia_bench);
`cpu_baseline` (rank 0, N = 1 only) times the unmodified reference (oracle/_ref, kind "reference") -- or the C
restatement (kind "port") when oracle/_ref is absent -- on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in KB, separate --pmc
# passes, collected at HEAD by scripts/profile_r03.sh on scripts/gpu_pmc_target.py = exactly these configurations).  MI355X_MICROARCH.md
# (HBM): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (>= 16 B per lane) -- "double it
# before comparing with a byte count"; other access widths are uncalibrated.  So: the histogram's row stream (64 B per lane) gets the
# x2 (`fetch_factor` 2), the point kernel's 32-byte gathers are reported as counted (factor 1, stated).  None when no summary is there.
# counters of the newest committed PMC evidence set (scripts/profile_r04.sh -> profiles/r04_pmc.json; the round-3 set as the fallback)
PMC_JSON = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json")) if os.path.exists(q)),
                os.path.join(ROOT, "profiles", "r06_pmc.json"))
PMC_NAME = "profiles/" + os.path.basename(PMC_JSON)


def _pmc():
    try:
        with open(PMC_JSON) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def pmc_is_current(source_file):
    """True while the kernel source the counters were collected on (sha256 recorded by scripts/summarize_prof.py pmc-json, `_meta`) is the one in this tree:
    a counter json that belongs to an older kernel is not reported as this run's traffic."""
    import hashlib
    pmc = _pmc()
    want = ((pmc or {}).get("_meta") or {}).get("kernel_sources_sha256", {}).get(source_file)
    if not want:
        return False
    try:
        with open(os.path.join(ROOT, "gpboost_amd", "csrc", source_file), "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest() == want
    except OSError:
        return False


def profiled_traffic_bytes(kernel_substr, fetch_factor=1.0, source_file=None):
    pmc = _pmc()
    if not pmc:
        return None
    if source_file is not None and not pmc_is_current(source_file):
        return None
    for name, c in pmc.items():
        if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return (fetch_factor * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    return None


def profiled_mfma_busy_cycles(kernel_substr):
    """Sum over the dispatches of the profiled run of SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's MFMA pipe was busy, summed over SIMDs), and
    the number of dispatches it is over."""
    pmc = _pmc()
    if not pmc:
        return None
    tot, n = 0.0, 0
    for name, c in pmc.items():         # all kernels that match (the update exists in several instantiations)
        if kernel_substr in name and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            tot += c["SQ_VALU_MFMA_BUSY_CYCLES"] * c.get("n", 1); n += c.get("n", 1)
    return (tot, n) if n else None


def laplace_kernel_rooflines(n4=100000, m4=30):
    """Per-kernel rooflines of the Vecchia-Laplace solvers from KERNEL durations (VERDICT r05 #1 i: the per-iteration figures of config4_vecchia_laplace divide host timers by
    iteration counts): mean durations of the committed rocprofv3 --kernel-trace runs of ONE config-4 evaluation per preconditioner (profiles/r06_a_trace_config4_*_summary.txt,
    scripts/gpu_r6_targets.py) against the algorithmic bytes of one launch.  Read at run time like `traffic`; None where no trace is committed."""
    import re
    fac = n4 * m4 * 16                       # the factor's {coefficient, source} records, once
    vec = n4 * 8
    kernels = (   # (substring of the kernel name, label, algorithmic bytes of one launch as a function of k)
        ("lap_sptrsv_sf_kernel<false, true", "single-vector triangular solve with B^T (barrier-free, ~118 wide levels)", lambda k: fac + 3 * vec),
        ("lap_sptrsv_sf_kernel<true, false", "single-vector triangular solve with B", lambda k: fac + 3 * vec),
        ("lap_sptrsv_sfw_kernel<false, true", "50-probe block solve with B^T (13 chunks of 4 columns share every entry load)", lambda k: fac + 2 * 52 * vec),
        ("lap_sptrsv_sfw_kernel<true, false", "50-probe block solve with B", lambda k: fac + 2 * 52 * vec),
        ("lap_tri_spmv_kernel<2, 4>", "B^T D^-1 B x on the 50-probe block", lambda k: 2 * fac + 2 * 52 * vec),
        ("pc_ltwx_kernel<4>", "L_k^T (W o X) on the 50-probe block: the n x k factor once per chunk of 4 columns", lambda k: 13 * (n4 * k * 8 + 5 * vec)),
        ("pc_combine_kernel<4>", "X - L_k x2 on the 50-probe block", lambda k: 13 * (n4 * k * 8 + 9 * vec)),
    )
    out = {}
    for tag, k in (("vadu", 0), ("pivchol", 50), ("fitc", 200), ("vecchia_response", 0)):
        path = os.path.join(ROOT, "profiles", ("r06_g_trace_config4_%s_summary.txt" if tag == "vecchia_response" else "r06_a_trace_config4_%s_summary.txt") % tag)
        if not os.path.exists(path):
            continue
        rows = []
        with open(path) as fh:
            txt = fh.read().split("\n")
        for sub, label, byt in kernels:
            for line in txt:
                if sub in line and "mean_us=" in line:
                    us = float(re.search(r"mean_us=\s*([0-9.]+)", line).group(1)); calls = int(re.search(r"calls=\s*([0-9]+)", line).group(1))
                    b = byt(k)
                    if b <= 0 or (k == 0 and sub.startswith("pc_")):
                        break
                    rows.append({"kernel": sub, "what": label, "calls_in_one_evaluation_plus_setup": calls, "mean_kernel_us": us, "algorithmic_bytes_per_launch": b,
                                 "achieved": b / (us * 1e-6) / 1e9, "unit": "GB/s", "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
                    break
        if rows:
            out[tag] = {"source": "%s (rocprofv3 --kernel-trace --stats; kernel durations, not host timers)" % os.path.relpath(path, ROOT), "kernels": rows}
    return out or None


def vif_kernel_rooflines(n=100000, m=30, k=200):
    """Per-kernel figures of the full-scale Vecchia x non-Gaussian evaluation / gradient (round 6) from KERNEL durations: the committed rocprofv3 --kernel-trace run of
    scripts/gpu_vif_non_gaussian.py at config 4's size (profiles/r06_zz_trace_vif_non_gaussian_config4_size_rocprofv3_summary.txt) against the algorithmic bytes of one launch."""
    import re
    path = os.path.join(ROOT, "profiles", "r06_zz_trace_vif_non_gaussian_config4_size_rocprofv3_summary.txt")
    if not os.path.exists(path):
        return None
    vec = n * 8; mat = n * k * 8; fac = n * m * 16; kq = (k + 1 + 7) // 8 * 8
    kernels = (
        ("pc_ltwx_kernel<1>", "L' (w o x): the n x k matrix once (low-rank part of Sigma x, the fitc preconditioner, the Woodbury quadratic forms)", mat + 2 * vec),
        ("pc_combine_kernel<1>", "x - L x2: the n x k matrix once", mat + 3 * vec),
        ("pc_ltwx_mfma_kernel<4>", "the same on the 50-probe block as a tall-skinny GEMM on v_mfma_f64_16x16x4_f64: the n x k matrix ONCE per launch (until round 6's last pass: once per chunk "
                                   "of 4 columns, 626 us); compute 2 n k 52 flops = 2.1 GFLOP", mat + vec + 13 * 4 * vec),
        ("pc_combine_mfma_kernel<4>", "X - L x2 on the 50-probe block, the same way (342 us before)", mat + vec + 2 * 13 * 4 * vec),
        ("lap_sptrsv_sf_kernel<false, true", "single-vector triangular solve with B^T of the residual factor (barrier-free; latency-bound dependency chain)", fac + 3 * vec),
        ("lap_sptrsv_sf_kernel<true, false", "single-vector triangular solve with B", fac + 3 * vec),
        ("vif_resid_factor_kernel", "residual-process factor: per point the whitened cross-covariances of its m + 1 points (gathered rows of V) + coordinates", n * (m + 1) * (kq * 8 + 32) + n * m * 12),
        ("pc_gram_kernel", "L' diag(w) L, k x k (gradient / vifdu; compute: n k (k + 1) flops)", mat + vec),
        ("pc_row_quad_tiled_kernel<true>", "row-wise quadratic forms with two k x k matrices (derivative of the preconditioner's diagonal; compute: 4 n k^2 flops)", 2 * mat + vec),
    )
    with open(path) as fh:
        txt = fh.read().split("\n")
    rows = []
    for sub, label, b in kernels:
        for line in txt:
            if sub in line and "mean_us=" in line:
                us = float(re.search(r"mean_us=\s*([0-9.]+)", line).group(1)); calls = int(re.search(r"calls=\s*([0-9]+)", line).group(1))
                rows.append({"kernel": sub, "what": label, "calls_in_the_traced_run": calls, "mean_kernel_us": us, "algorithmic_bytes_per_launch": b,
                             "achieved": b / (us * 1e-6) / 1e9, "unit": "GB/s", "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
                break
    return {"source": "%s (rocprofv3 --kernel-trace --stats: three evaluations + a five-iteration lbfgs fit; kernel durations, not host timers)" % os.path.relpath(path, ROOT),
            "kernels": rows} if rows else None


HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 x 2.4 GHz (vector == matrix fp64 rate on gfx950)


def algorithmic_bytes_per_point(m, d):
    """neighbour indices + gathered coords of the point and its m neighbours + gathered y."""
    return 4 * m + 8 * d * (m + 1) + 8 * (m + 1)


def algorithmic_flops_per_point(m, d, cov_type):
    """m^3/3 + 2 m^2 + m(m+1)/2 (3d + c_k) + 4m, each exp/sqrt/div counted as 1."""
    ck = {0: 3, 1: 5, 2: 8}[cov_type]
    return m ** 3 / 3.0 + 2 * m * m + (m * (m + 1) / 2.0) * (3 * d + ck) + 4 * m


def _omp_set_num_threads(k):
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(k))
    except OSError:
        pass


def cpu_baseline(coords, y, cov_function, shape, m, cov_pars, n_full):
    """The reference's CPU path (oracle/_ref: the unmodified reference's GPB_EvalNegLogLikelihood, kind "reference"; the C restatement,
    kind "port", when oracle/_ref is absent) timed DIRECTLY at the metric's n (no scaling) at several OpenMP thread counts; the best is
    reported with its count.  Bounded: one model creation + one evaluation per thread count (about 20-40 s of wall time in total)."""
    n_s = len(y)
    hw = os.cpu_count() or 1
    counts = sorted({c for c in (8, 32, 64, hw) if c <= hw})
    from oracle import refdrv
    if refdrv.available():
        mdl = refdrv.RefCAPIModel(coords, cov_function, shape, m, "random", 1, threads=hw)
        f = lambda cp: mdl.neg_log_likelihood(cp, y)
        kind = "reference"
    else:
        from oracle import orc
        _omp_set_num_threads(hw)
        setup = orc.vecchia_setup(coords, m, "random", 1)
        f = lambda cp: orc.gp_nll(coords, y, cp, cov_function, shape, m, setup=setup)
        kind = "port"
    # warm-up, then >= 3 timed evaluations with perturbed parameters, median.  Leg 1 (doubles as the warm-up of the
    # pages and of the OpenMP pools): ONE evaluation per thread count picks the best count; leg 2: three more timed evaluations at that count.
    scan = {}
    budget_t0 = time.perf_counter()
    for k, c in enumerate(reversed(counts)):     # most threads first
        if time.perf_counter() - budget_t0 > 60.0 and scan:
            break
        _omp_set_num_threads(c)
        cp = cov_pars * (1.0 + 0.01 * (k + 1))
        t0 = time.perf_counter(); f(cp); scan[c] = time.perf_counter() - t0
    best = min(scan, key=scan.get)
    _omp_set_num_threads(best)
    timed = []
    for k in range(3):
        cp = cov_pars * (1.0 + 0.003 * (k + 1))
        t0 = time.perf_counter(); f(cp); timed.append(time.perf_counter() - t0)
        if time.perf_counter() - budget_t0 > 120.0:
            break
    med = float(np.median(timed))
    return {"value": 1.0 / med, "unit": "evals/s", "cores": best, "kind": kind, "seconds_per_eval_median": med, "timed_evals": len(timed),
            "sample": "n=%d (the metric's size, no scaling), m=%d; thread-count scan, one evaluation each (also the warm-up): %s; then %d timed "
                      "evaluations with perturbed parameters at the best count (%d of %d hardware threads): %s s, median %.3f s"
                      % (n_s, m, ", ".join("%d thr %.2f s" % (c, scan[c]) for c in sorted(scan)), len(timed), best, hw,
                         " / ".join("%.3f" % t for t in timed), med)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--m", type=int, default=30)
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--cov", default="exponential", choices=["exponential", "matern_1.5", "matern_2.5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the metric's line: skip the other configurations' measurements reported next to it")
    # (the self-launcher below hands the ranks their arguments through GPB_BENCH_ARGV: torch.distributed.run's own parser would reject an
    #  abbreviation-ambiguous script option such as --n before it ever reaches the script)
    argv = json.loads(os.environ["GPB_BENCH_ARGV"]) if (len(sys.argv) == 1 and "GPB_BENCH_ARGV" in os.environ) else sys.argv[1:]
    args = ap.parse_args(argv)

    # --gpus N without a launcher (WORLD_SIZE unset): this process becomes the launcher of its own N ranks -- the same command line the driver
    # uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...), so that
    # `python bench.py --gpus N` and the torchrun form walk the same code.  A line is only ever printed with n_gpus == --gpus (checked below).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)]
        os.environ["GPB_BENCH_ARGV"] = json.dumps(sys.argv[1:])
        os.execv(sys.executable, cmd)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GPB_BENCH_FORCE_DIST=1 exercises the multi-GPU code path (RCCL init, shared stream, all-reduce) with one rank
    distributed = world > 1 or os.environ.get("GPB_BENCH_FORCE_DIST", "0") == "1"
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: no line is printed whose n_gpus differs from --gpus "
                         "(run `python bench.py --gpus %d`, which starts its own ranks)" % (args.gpus, world, world))

    # STDOUT carries the one JSON line and nothing else: libraries announce themselves there (RCCL prints its version banner on the first
    # communicator, gloo its connections), so the descriptor is parked and fd 1 points at stderr until rank 0 writes the line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import gpboost_amd
    from gpboost_amd import parallel, shim
    torch = dist = None
    rehearsal = False
    cdev = "cpu"                 # where the bootstrap's small tensors live: the GPU under RCCL, the host under gloo (rehearsal)
    if distributed:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise RuntimeError("bench.py: no GPU visible")
        # REHEARSAL (one-GPU lease): fewer devices than ranks -> every rank uses device 0, the process group is gloo, RCCL (one device per
        # rank) stays down and the shard sums travel through the node-local mailbox -- after the bootstrap exactly the code of a real
        # N-device launch.  The line says so ("rehearsal": true, the metric's name too): its value is NOT a scaling number, the N shard
        # kernels time-share one device.  GPB_BENCH_REHEARSAL=1 forces it, =0 forbids it.
        want = os.environ.get("GPB_BENCH_REHEARSAL", "")
        rehearsal = (want == "1") or (ndev < world and want != "0")
        if ndev < world and not rehearsal:
            raise RuntimeError("bench.py --gpus %d: only %d device(s) visible and GPB_BENCH_REHEARSAL=0" % (world, ndev))
        if rehearsal:
            torch.cuda.set_device(0)
            gpboost_amd.set_device(0)
            dist.init_process_group("gloo")
            dist.barrier()
            if rank == 0:
                print("bench.py: REHEARSAL -- %d ranks share device 0 (%d device(s) visible); gloo bootstrap, mailbox for the sums" % (world, ndev), file=sys.stderr)
        else:
            torch.cuda.set_device(local_rank)
            gpboost_amd.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            cdev = "cuda"
    if not distributed:
        gpboost_amd.set_device(0)

    cov_function, shape = {"exponential": ("exponential", 0.5), "matern_1.5": ("matern", 1.5),
                           "matern_2.5": ("matern", 2.5)}[args.cov]
    ct = {0.5: 0, 1.5: 1, 2.5: 2}[shape]
    n, m, d = args.n, args.m, args.d
    rng = np.random.default_rng(1)                     # BASELINE.md section 2 inputs
    coords = rng.uniform(size=(n, d))
    y = rng.standard_normal(n)
    cov_pars = np.array([0.1, 1.0, 0.1])               # (sigma2, sigma1_2, rho)
    sigma2 = cov_pars[0]
    var0 = cov_pars[1] / cov_pars[0]
    a0 = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cov_pars[2]

    t0 = time.perf_counter()
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=cov_function, cov_fct_shape=shape, gp_approx="vecchia",
                              num_neighbors=m, vecchia_ordering="random", seed=1)
    t_setup = time.perf_counter() - t0
    perm, _ = mdl.vecchia_structure()
    st = shim.VecchiaState.from_handle(mdl.vecchia_handle(), n, d, m)
    # y is uploaded ONCE (this call; GPB_EvalNegLogLikelihood with a host pointer permutes it to the Vecchia order and copies it to
    # HBM) and stays resident: every timed evaluation passes y_data = NULL (re_model_template.h:2905-2921)
    mdl.neg_log_likelihood(cov_pars, y)
    # (model set-up, like the neighbour search: the library builds the spatially sorted copy of the records its gathers read at the third evaluation on a
    #  neighbour table -- a host sort, ~0.1 s per million points; DESIGN.md section 4.1.  Two more untimed evaluations put it before ANY warm-up count.)
    mdl.neg_log_likelihood(cov_pars)
    mdl.neg_log_likelihood(cov_pars)
    i0, i1 = parallel.shard_range(n, rank, world)
    st.set_shard(i0, i1)

    tdev = None
    native_rccl = False
    if distributed:
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        tdev = torch.zeros(3, dtype=torch.float64, device="cuda")

        def all_min(flag):          # every rank learns whether EVERY rank succeeded
            t = torch.tensor([int(flag)], dtype=torch.int32, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())

        def bcast_bytes(payload, nbytes):     # rank 0's bytes to every rank
            t = torch.zeros(nbytes, dtype=torch.uint8, device=cdev)
            if rank == 0:
                t.copy_(torch.tensor(list(payload.ljust(nbytes, b"\0")), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            return bytes(t.cpu().tolist())

        # In-library RCCL reduction (point kernel -> reduction -> ncclAllReduce -> host on ONE stream, one sync per
        # evaluation) unless GPB_BENCH_TORCH_ALLREDUCE=1; the unique id travels over the torch process group.  Every rank
        # reports whether its communicator came up; if any did not, all ranks use torch.distributed.all_reduce instead.
        # (Rehearsal: RCCL refuses two ranks on one device -- it stays down, rccl_ranks = 0, and the mailbox below is the only transport.)
        ok = 0
        if not rehearsal and os.environ.get("GPB_BENCH_TORCH_ALLREDUCE", "0") != "1":
            try:
                st.comm_init(bcast_bytes(bytes(shim.comm_unique_id()) if rank == 0 else b"", 128), rank, world)
                ok = 1
            except Exception as e:   # noqa: BLE001
                print("rank %d: native RCCL communicator failed (%s); falling back to torch all_reduce" % (rank, e), file=sys.stderr)
        native_rccl = (not rehearsal) and all_min(ok) == 1
        # The 3 shard sums of an evaluation travel through the node-local shared-memory MAILBOX: every rank's finisher
        # workgroup stores its sums into its slot, every host polls all slots -- no collective launch per evaluation.  RCCL stays up for the big
        # messages.  GPB_BENCH_NO_MAILBOX=1 keeps ncclAllReduce for the sums (A/B).
        mok = 0
        if os.environ.get("GPB_BENCH_NO_MAILBOX", "0") != "1":      # (the mailbox does not need RCCL: a job whose communicator failed still runs the product's sums)
            try:
                nm = bcast_bytes(shim.mailbox_create(world) if rank == 0 else b"", 64)
                st.mailbox_attach(nm.rstrip(b"\0"), rank, world)
                mok = 1
            except Exception as e:   # noqa: BLE001
                print("rank %d: mailbox failed (%s)" % (rank, e), file=sys.stderr)
            if all_min(mok) != 1:
                if mok:
                    st.mailbox_detach()
                mok = 0
            if mok:
                # one evaluation through the mailbox, checked before anything is timed: the job's sums must equal the ranks' shard sums added in rank
                # order (all-gathered over torch) bit for bit -- the mailbox adds them in exactly that order.  Any rank that fails (a time-out of the
                # bounded poll included) sends every rank back to ncclAllReduce for the sums.
                try:
                    got = np.asarray(st.nll_terms_allreduce(ct, var0, a0))[:3]
                    loc = torch.tensor(np.asarray(st.nll_terms(ct, var0, a0))[:3], dtype=torch.float64, device=cdev)
                    parts = [torch.zeros(3, dtype=torch.float64, device=cdev) for _ in range(world)]
                    dist.all_gather(parts, loc)
                    want = np.zeros(3)
                    for pr in parts:
                        want = want + pr.cpu().numpy()
                    if not np.array_equal(got, want):
                        raise RuntimeError("mailbox sums %r differ from the rank-ordered sums %r" % (got.tolist(), want.tolist()))
                except Exception as e:   # noqa: BLE001
                    print("rank %d: mailbox check failed (%s); the sums go through ncclAllReduce" % (rank, e), file=sys.stderr)
                    mok = 0
                if all_min(mok) != 1:
                    st.mailbox_detach()
                    mok = 0
        use_mailbox = bool(mok)
        # Under a multi-rank launch a silent fallback would report a number for a path that is not the product's: fail loudly instead
        # (GPB_BENCH_TORCH_ALLREDUCE=1 asks for the torch fallback explicitly).
        if world > 1 and not native_rccl and not use_mailbox and os.environ.get("GPB_BENCH_TORCH_ALLREDUCE", "0") != "1":
            raise RuntimeError("bench.py --gpus %d: neither the in-library RCCL communicator nor the mailbox came up on every rank (see stderr); "
                               "GPB_BENCH_TORCH_ALLREDUCE=1 selects the torch.distributed fallback explicitly" % world)
        # A real multi-device launch must have the in-library RCCL communicator up on every rank: it carries the big messages of the N > 1
        # path and is the A/B transport of the sums -- a line with rccl_ranks != N is not the job that was asked for.  GPB_BENCH_ALLOW_NO_RCCL=1 lifts it.
        if world > 1 and not native_rccl and not rehearsal and os.environ.get("GPB_BENCH_ALLOW_NO_RCCL", "0") != "1" and os.environ.get("GPB_BENCH_TORCH_ALLREDUCE", "0") != "1":
            raise RuntimeError("bench.py --gpus %d: the in-library RCCL communicator did not come up on every rank (see stderr): no line is printed with "
                               "rccl_ranks != %d (GPB_BENCH_ALLOW_NO_RCCL=1 prints the mailbox-only job anyway)" % (world, world))
        if rehearsal and not use_mailbox:
            raise RuntimeError("bench.py --gpus %d (rehearsal): the mailbox did not come up on every rank (see stderr)" % world)
    else:
        use_mailbox = False
    native = native_rccl or use_mailbox      # the library reduces the shard sums inside GPB_EvalNegLogLikelihood (RCCL or the mailbox)

    def cov_pars_of(k):
        # covariance parameters change every evaluation (perturbed by <= 1 %): nothing is reusable between steps
        return np.array([sigma2, cov_pars[1] * (1.0 + 0.002 * ((k % 11) - 5)), cov_pars[2] / (1.0 + 0.002 * ((k % 7) - 3))])

    # THE metric: wall time of GPB_EvalNegLogLikelihood(handle, y_data = NULL, cov_pars, NULL, &negll) -- on a sharded
    # handle the library all-reduces the 3 shard sums over RCCL inside the call, every rank gets the job's value.  The call is made the
    # way the reference's package makes it (ctypes, basic.py:5640-5700) with the arguments marshalled ahead of the timed region: the
    # parameter sets of all steps exist as arrays before the clock starts, a step = ONE foreign call + reading the returned double.
    import ctypes
    from gpboost_amd.basic import _lib, _safe_call
    c_eval = _lib().GPB_EvalNegLogLikelihood
    c_eval.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    c_eval.restype = ctypes.c_int
    c_negll = ctypes.c_double(0.0)
    c_negll_ref = ctypes.byref(c_negll)
    c_handle = mdl.handle
    n_sets = 77                                   # lcm(11, 7): every parameter set cov_pars_of produces
    cp_sets = [np.ascontiguousarray(cov_pars_of(k)) for k in range(n_sets)]
    cp_ptrs = [cp.ctypes.data for cp in cp_sets]

    def one_eval(k):
        if distributed and not native:      # fallback only: shard terms on the device + torch.distributed all-reduce
            cpk = cov_pars_of(k)
            st.nll_terms_dev(ct, cpk[1] / cpk[0], a0 * cov_pars[2] / cpk[2], tdev.data_ptr())
            dist.all_reduce(tdev)
            t = tdev.cpu().numpy()
            return parallel.nll_from_terms(n, t[0], t[1], sigma2)
        rc = c_eval(c_handle, None, cp_ptrs[k % n_sets], None, c_negll_ref)
        if rc != 0:
            _safe_call(rc)
        return c_negll.value

    def sync():
        if distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            st.sync()

    # untimed pre-warm (set-up, not part of W, stated in config.prewarm_evals): the first few hundred milliseconds after the
    # neighbour search run at a lower clock / colder caches; the W warm-up steps the contract asks for follow it
    PREWARM = 200            # fixed count: every rank must issue the same number of collectives
    for _ in range(PREWARM):
        one_eval(0)
    for k in range(args.warmup):
        one_eval(k)
    sync()
    # the dominant kernel is timed INSIDE the timed loop: a HIP event pair around every point-kernel launch on the handle's stream
    # (gpb_hip_vecchia_timing; read after the loop), so kernel_ms <= ms_per_step by construction
    st.timing(True)
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = one_eval(args.warmup + k)
    sync()
    dt = time.perf_counter() - t0
    timed_launches, ms_kernel_inloop = st.timing(False)
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # the same call through the package-style wrapper (GPModel.neg_log_likelihood: argument checks + numpy marshalling per call), untimed
    # steps first so that both loops run at the same clocks; reported in config, never as `value`
    wrapper_ms = None
    if not distributed or native:
        for k in range(5):
            mdl.neg_log_likelihood(cov_pars_of(k))
        sync()
        tw0 = time.perf_counter()
        for k in range(args.steps):
            mdl.neg_log_likelihood(cov_pars_of(args.warmup + k))
        sync()
        wrapper_ms = (time.perf_counter() - tw0) / args.steps * 1e3

    # A/B inside the same run (real multi-device launches only): the same timed call with the sums through ncclAllReduce (+ publish kernel) instead of
    # the mailbox -- north_star names "RCCL all-reduce of log-likelihood scalars"; both rates are in the line, `value` is the mailbox's (the default)
    ab_rccl = None
    if distributed and native_rccl and use_mailbox and os.environ.get("GPB_BENCH_NO_AB", "0") != "1":
        mbox_name_ranks = st.mailbox_info()[1]
        st.mailbox_detach()
        try:
            for k in range(max(args.warmup, 3)):
                one_eval(k)
            sync()
            ta0 = time.perf_counter()
            for k in range(args.steps):
                one_eval(args.warmup + k)
            sync()
            dta = time.perf_counter() - ta0
            tta = torch.tensor([dta], dtype=torch.float64, device=cdev)
            dist.all_reduce(tta, op=dist.ReduceOp.MAX)
            dta = float(tta.item())
            ab_rccl = {"ms_per_step": dta / args.steps * 1e3, "evals_per_s": args.steps / dta, "rccl_ranks": st.comm_info()[1],
                       "call": "the same GPB_EvalNegLogLikelihood loop with the mailbox detached: point kernel -> ncclAllReduce(3 fp64) -> publish kernel -> host poll"}
        finally:
            # back to the mailbox for everything measured after this (a fresh segment: the old one's generations ended with the detach)
            nm = bcast_bytes(shim.mailbox_create(world) if rank == 0 else b"", 64)
            st.mailbox_attach(nm.rstrip(b"\0"), rank, world)
        assert st.mailbox_info()[1] == mbox_name_ranks

    # batched entry point (GPB_HIP_EvalNegLogLikelihoodBatch): K parameter sets per call, ONE synchronisation and ONE all-reduce of 3 K
    # doubles -- what an optimiser's line search / a grid of trial points would use; reported next to the metric, never as `value`
    batched = None
    if not distributed or native:
        Kb = 32
        cps = np.stack([cov_pars_of(k) for k in range(Kb)])
        mdl.neg_log_likelihood_batch(cps)
        sync()
        tb0 = time.perf_counter()
        for _ in range(4):
            mdl.neg_log_likelihood_batch(cps)
        sync()
        dtb = time.perf_counter() - tb0
        if distributed:
            ttb = torch.tensor([dtb], dtype=torch.float64, device=cdev)
            dist.all_reduce(ttb, op=dist.ReduceOp.MAX)
            dtb = float(ttb.item())
        batched = {"K": Kb, "evals_per_s": 4 * Kb / dtb, "ms_per_eval": dtb / (4 * Kb) * 1e3,
                   "call": "GPB_HIP_EvalNegLogLikelihoodBatch: one synchronisation and one all-reduce per K evaluations"}

    # dominant kernel, measured with HIP events on its own stream (this rank's shard)
    ms_total, ms_kernel_sep, _ = st.bench(shim.MODE_NLL, ct, var0, a0, 1, max(3, min(args.steps, 20)))      # a separate loop of launches (cross-check)
    ms_gtotal, ms_gkernel, _ = st.bench(shim.MODE_GRAD, ct, var0, a0, 1, 3)
    ms_kernel = ms_kernel_inloop if (timed_launches >= args.steps and ms_kernel_inloop > 0) else ms_kernel_sep
    npts = i1 - i0
    bytes_launch = npts * algorithmic_bytes_per_point(m, d)
    flops_launch = npts * algorithmic_flops_per_point(m, d, ct)
    achieved_gbs = bytes_launch / (ms_kernel * 1e-3) / 1e9
    achieved_tflops = flops_launch / (ms_kernel * 1e-3) / 1e12

    # <MT, COV, D3, MODE_NLL, WT = no sample weights>; the PMC passes were collected at n = 1e6 on one GPU: no figure for any other launch
    traffic = (profiled_traffic_bytes("vecchia_point_kernel<%d, %d, %s, 0, false>" % (m, ct, "true" if d == 3 else "false"), source_file="vecchia_kernels.hip")
               if (n, world) == (1000000, 1) else None)
    rccl_ranks = st.comm_info()[1] if native_rccl else 0
    mailbox_ranks = st.mailbox_info()[1] if use_mailbox else 0
    # per-rank view (N > 1): every rank's in-loop kernel time and shard size -- min / max over the ranks is the skew the N = 8 budget of DESIGN.md
    # section 5 leaves open; and what every rank's transport says about itself (the line is refused if they disagree with the launch)
    per_rank = None
    if distributed:
        mine = torch.tensor([ms_kernel, float(npts), float(rccl_ranks), float(mailbox_ranks)], dtype=torch.float64, device=cdev)
        allr = [torch.zeros(4, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = np.stack([t.cpu().numpy() for t in allr])
        per_rank = {"kernel_ms": [round(float(v), 5) for v in allr[:, 0]], "kernel_ms_min": float(allr[:, 0].min()), "kernel_ms_max": float(allr[:, 0].max()),
                    "shard_points": [int(v) for v in allr[:, 1]], "rccl_ranks_seen": [int(v) for v in allr[:, 2]], "mailbox_ranks_seen": [int(v) for v in allr[:, 3]]}
        if world > 1:
            seen = allr[:, 3] if use_mailbox else allr[:, 2]
            if native and not np.all(seen == world):
                raise RuntimeError("bench.py --gpus %d: the ranks' transport reports %r ranks -- no line is printed for a job that is not the one asked for" % (world, seen.tolist()))
            if int(allr[:, 1].sum()) != n:
                raise RuntimeError("bench.py --gpus %d: the shards cover %d of %d points" % (world, int(allr[:, 1].sum()), n))
    if rank == 0:
        out = {
            "metric": "neg-log-lik evals/sec, n=%d Vecchia(m=%d) fp64" % (n, m) + (" [REHEARSAL: %d ranks time-share ONE device -- not a scaling number]" % world if rehearsal else ""),
            "value": args.steps / dt,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "Vecchia GP Gaussian nll, n=%d, d=%d, %s, m=%d, vecchia_ordering=random" % (n, d, args.cov, m),
                       "timed_call": ("GPB_EvalNegLogLikelihood(handle, y_data=NULL, cov_pars, fixed_effects=NULL, &negll) through ctypes: y resident in HBM, "
                                      "parameters in, value out" if (not distributed or native) else "shard terms + torch.distributed all_reduce (fallback path)"),
                       "shard_points_per_gpu": npts, "parallelism": "points sharded x%d, sum of 3 fp64 over the ranks (%s)" % (world, ("node-local shared-memory mailbox, %d ranks (RCCL up with %d ranks for the big messages)" % (mailbox_ranks, rccl_ranks)) if use_mailbox else ("in-library RCCL, %d ranks" % rccl_ranks if native_rccl else ("torch.distributed nccl" if distributed else "single GPU"))) + (" -- REHEARSAL on one device" if rehearsal else ""),
                       "rccl_ranks": rccl_ranks, "mailbox_ranks": mailbox_ranks, "prewarm_evals": PREWARM, "rehearsal": rehearsal, "per_rank": per_rank, "ab_sums_through_ncclAllReduce": ab_rccl,
                       "process_group": ("gloo (rehearsal bootstrap)" if rehearsal else "nccl (RCCL)") if distributed else None,
                       "kernel_ms_source": "HIP events around every point-kernel launch INSIDE the timed loop (%d launches); separate loop of launches: %.4f ms" % (timed_launches, ms_kernel_sep),
                       "setup_s_model_creation_incl_device_neighbor_search": round(t_setup, 3),
                       "last_negll": last, "batched": batched,
                       # what one evaluation costs outside the point kernel (launch, the in-kernel final sums, pinned-memory hand-over, the
                       # all-reduce for N > 1, the foreign call): ms_per_step - kernel_ms, both measured in this run
                       "overhead_us": round((dt / args.steps * 1e3 - ms_kernel) * 1e3, 2),
                       "ms_per_step_through_python_wrapper": wrapper_ms,
                       "grad_eval_ms_kernel": round(ms_gkernel, 4), "grad_over_nll_kernel_time": round(ms_gkernel / ms_kernel, 3)},
            # BASELINE.json's metric asks for "% HBM roofline": the primary object is the HBM view of the dominant kernel (algorithmic gather
            # bytes of SURVEY.md 8d / HIP-event kernel time / 8 TB/s).  The kernel is NOT HBM-bound -- it is bound by fp64 VALU issue
            # (DESIGN.md 4.1); that view is `roofline_fp64_valu` (no MFMA is issued; vector fp64 peak 78.6 TFLOP/s).
            # `traffic` = FETCH_SIZE + WRITE_SIZE of the same kernel from the PMC passes committed under profiles/ (read at run time).
            "roofline": {"bound": "hbm", "kernel": "vecchia_point_kernel<MODE_NLL>", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": ms_kernel, "algorithmic_bytes_per_launch": bytes_launch,
                         "traffic_source": (PMC_NAME + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 32-byte gathers: counted as reported, no x2; the json's recorded sha256 of vecchia_kernels.hip equals this tree's)") if traffic else
                                           ("null: " + PMC_NAME + " was collected on another version of vecchia_kernels.hip (sha256 mismatch) -- rerun scripts/gpu_run.sh <tag> pmc" if (n, world) == (1000000, 1) else None),
                         "note": "binding resource is fp64 VALU issue, not HBM: see roofline_fp64_valu; 0.864 GB of gathers per launch, the 32 MB record array lives in L2 / Infinity Cache"},
            "roofline_fp64_valu": {"bound": "fp64 vector ALU issue (no MFMA in this kernel)", "kernel": "vecchia_point_kernel<MODE_NLL>", "achieved": achieved_tflops,
                                   "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tflops / FP64_PEAK_TFLOPS, "kernel_ms": ms_kernel,
                                   "algorithmic_flops_per_launch": flops_launch,
                                   "note": "exp / sqrt / division counted as ONE flop each (SURVEY.md 8d); instruction mix: profiles/r04_point_kernel_instruction_mix.txt"},
        }
        if world == 1 and not args.no_extras:
            # the one HBM-bound kernel of the path: dense covariance assembly (exact GP), measured live
            try:
                ne = 16384
                ex = shim.ExactState(coords[:ne]); ex.set_y(y[:ne])
                ex.nll_terms(ct, var0, a0)
                _, _, ms3 = ex.nll_terms(ct, var0, a0)
                nt = (ne + 127) // 128
                wbytes = nt * (nt + 1) // 2 * 128 * 128 * 8          # lower 128x128 tiles actually written
                out["roofline_cov_assembly"] = {
                    "bound": "hbm", "kernel": "dense_cov_lower_kernel", "achieved": wbytes / (ms3[0] * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": wbytes / (ms3[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel_ms": float(ms3[0]), "algorithmic_bytes_per_launch": wbytes,
                    "workload": "exact GP covariance assembly, n=%d (lower triangle, 8 B written per Matern evaluation)" % ne,
                    "dense_cholesky_ms": float(ms3[1]), "dense_cholesky_tflops": ne ** 3 / 3.0 / (ms3[1] * 1e-3) / 1e12,
                    "fp64_mfma_peak_tflops": FP64_PEAK_TFLOPS,
                    "dense_cholesky_frac_of_fp64_mfma_peak": ne ** 3 / 3.0 / (ms3[1] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
                # MFMA utilisation of the panel GEMMs from the counters (profiles/r03_pmc.json: the same n = 16384 factorisation, three of them
                # in the profiled run): SQ_VALU_MFMA_BUSY_CYCLES summed over the syrk_mfma_kernel dispatches of ONE factorisation / (SIMD-cycles
                # of the factorisation measured here: ms x 2.4 GHz x 1024 SIMDs)
                mb = profiled_mfma_busy_cycles("syrk_mfma")
                if mb:
                    out["roofline_cov_assembly"]["mfma_busy_cycles_per_factorisation"] = mb[0] / 3.0
                    out["roofline_cov_assembly"]["mfma_utilisation_pmc"] = mb[0] / 3.0 / (ms3[1] * 1e-3 * 2.4e9 * 1024)
                    out["roofline_cov_assembly"]["mfma_utilisation_source"] = PMC_NAME + ": SQ_VALU_MFMA_BUSY_CYCLES (64 cycles per v_mfma_f64_16x16x4) of the syrk_mfma_* kernels, %d dispatches = 3 factorisations" % mb[1]
                ex.close()
            except Exception as e:
                out["roofline_cov_assembly"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                # BASELINE config 1: exact GP, n = 2000, 2D, Matern-1.5 -- one likelihood evaluation through the shim (assembly + factorisation
                # with y as an extra row + log-det / quadratic form)
                rng1 = np.random.default_rng(2000)
                c1 = rng1.uniform(size=(2000, 2)); ex1 = shim.ExactState(c1); ex1.set_y(rng1.standard_normal(2000))
                for _ in range(3):
                    ex1.nll_terms(1, 1.0, np.sqrt(3.0) / 0.1)
                t1 = []
                for _ in range(20):
                    tt = time.perf_counter(); ex1.nll_terms(1, 1.0, np.sqrt(3.0) / 0.1); t1.append((time.perf_counter() - tt) * 1e3)
                out["config1_exact_gp_n2000"] = {"ms_per_evaluation": float(np.median(t1)), "workload": "exact GP, n=2000, 2D, Matern-1.5, Gaussian likelihood: one negative log-likelihood evaluation (gpb_hip_exact_nll_terms)"}
                ex1.close()
            except Exception as e:
                out["config1_exact_gp_n2000"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_extras:
            # LightGBM feature-histogram build at a size where GB/s means something (8d: n = 1e7 rows, F = 50,
            # 255 bins, constant hessian): algorithmic bytes = rows * (F + 8 + 4) in + F * bins * 16 out
            try:
                nh, Fh, nbh = 10000000, 50, 255
                rngh = np.random.default_rng(2)
                binsh = rngh.integers(0, nbh, size=(Fh, nh), dtype=np.uint8)
                boh = (np.arange(Fh + 1) * nbh).astype(np.int32)
                hb = shim.HistBuilder(binsh, boh)
                hb.set_gradients(rngh.standard_normal(nh), None)
                del binsh
                hb.bench(None, 1.0, 2)
                ms_h = hb.bench(None, 1.0, 10)
                hbytes = nh * (Fh + 8 + 4) + Fh * nbh * 16
                out["roofline_histogram"] = {
                    "bound": "hbm", "kernel": "hist_build_rows_kernel + hist_reduce_kernel", "achieved": hbytes / (ms_h * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbytes / (ms_h * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms_h,
                    "algorithmic_bytes_per_launch": hbytes, "traffic": profiled_traffic_bytes("hist_build_rows_kernel<false", fetch_factor=2.0, source_file="hist_kernels.hip"),
                    "traffic_source": PMC_NAME + ": 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction for wide coalesced streams); round 6: the root pass streams the COMPACT copy of the rows (52 bytes per row at F = 50; the padded 64-byte rows cost 1.24x the algorithmic bytes)",
                    "workload": "root-leaf histogram, n=%d rows, F=%d features, %d bins, constant hessian (counts exact)" % (nh, Fh, nbh),
                    "note": "fixed-point sums (one 64-bit LDS atomic per row and feature, count packed in, bank-conflict-free layout, a whole "
                            "64-byte row per lane): bit-reproducible, counts exact; see DESIGN.md 4.4"}
                hb.close()
            except Exception as e:
                out["roofline_histogram"] = {"error": "%s: %s" % (type(e).__name__, e)}
            # BASELINE config 4: Vecchia-Laplace, Bernoulli-logit, n = 1e5, m = 30 -- seconds per evaluation
            try:
                n4 = 100000
                rng4 = np.random.default_rng(1)
                c4 = rng4.uniform(size=(n4, 2))
                y4 = (rng4.uniform(size=n4) < 0.5).astype(np.float64)
                m4 = gpboost_amd.GPModel(likelihood="bernoulli_logit", gp_coords=c4, cov_function="exponential", gp_approx="vecchia",
                                         num_neighbors=30, vecchia_ordering="random", seed=1)
                m4.neg_log_likelihood(np.array([1.0, 0.1]), y4)          # builds the level schedules and probe vectors
                t4 = time.perf_counter()
                v4 = m4.neg_log_likelihood(np.array([1.01, 0.1]), y4)
                s4 = time.perf_counter() - t4
                i4 = m4.laplace_info()
                out["config4_vecchia_laplace"] = {
                    "workload": "Bernoulli-logit Vecchia-Laplace nll (Newton + vadu-CG + SLQ, 50 probes), n=%d, m=30, exponential" % n4,
                    "s_per_eval": s4, "negll": v4, "newton_it": i4["newton_it"], "cg_it": i4["cg_it"], "lanczos_it": i4["lanczos_it"],
                    "ms_mode_finding": i4["ms_mode"], "ms_logdet": i4["ms_logdet"],
                    # one preconditioned-CG iteration of the mode finding streams the factor four times (B and B' products of Sigma^-1 + W,
                    # B' and B triangular solves of the preconditioner: 16-byte {coefficient, source} entries + one 16-byte slot record per row
                    # and pass) and ~10 n-vectors: what it would cost at the HBM rate against what the level-scheduled solves take
                    "roofline_cg_iteration": (lambda byt, ms: {
                        "bound": "hbm", "kernel": "lap_tri_spmv_kernel x2 + triangular solves (dense block of the ~270 narrow levels: lap_dense_matvec_kernel, 268 MB at ~5.6 TB/s; lap_sptrsv_sf_kernel: ONE barrier-free launch per solve for the ~118 wide levels each way) + cg_* vector kernels",
                        "algorithmic_bytes_per_iteration": byt, "ms_per_iteration": ms, "achieved": byt / (ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "latency-bound, not bandwidth-bound: the two triangular solves are dependency chains of ~118 levels, one visibility round trip (~2.4 us) per level whatever the level's size (DESIGN.md 4.6: the floor of this formulation is ~0.33 s per evaluation); algorithmic bytes do not count the dense inverse blocks (2 x 268 MB per iteration), which replace ~540 dependent level steps"})(
                        4 * n4 * (30 * 16 + 16) + 10 * n4 * 8, i4["ms_mode"] / max(i4["cg_it"], 1)),
                    # one Lanczos / block-CG iteration of the log-determinant: the factor streamed four times for ALL 50 probe vectors (13 chunks of four
                    # columns share every entry load) + ~10 n-vectors per probe
                    "roofline_logdet_iteration": (lambda byt, ms: {
                        "bound": "hbm", "kernel": "lap_sptrsv_sfw_kernel x2 (round 4: the 50-probe block barrier-free, one launch per solve, one wavefront per (row, unit of four chunks)) + lap_dense_gemm4_kernel x2 + lap_tri_spmv_kernel<.,4> x2 + cg_* block kernels",
                        "algorithmic_bytes_per_iteration": byt, "ms_per_iteration": ms, "achieved": byt / (ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})(
                        4 * n4 * (30 * 16 + 16) + 10 * n4 * 8 * 52, i4["ms_logdet"] / max(i4["lanczos_it"], 1)),
                    }
                # round 5: the same evaluation with the reference's other built preconditioners, cg_preconditioner_type = "pivoted_cholesky" (rank 50) / "fitc" (200 inducing points): the solves in the
                # (W^-1 + Sigma) form, fewer (latency-bound) iterations -- another algorithm of the reference, not a faster kernel
                # round 6: + "vecchia_response" (P = the Vecchia factor of W^-1 + Sigma, renewed per Newton step by one point-kernel launch; evaluation only, as in the reference)
                for pcname in ("pivoted_cholesky", "fitc", "vecchia_response"):
                    try:
                        m4.set_optim_params({"cg_preconditioner_type": pcname})
                        m4.neg_log_likelihood(np.array([1.0, 0.1]), y4)
                        t4 = time.perf_counter()
                        v4p = m4.neg_log_likelihood(np.array([1.01, 0.1]), y4)
                        s4p = time.perf_counter() - t4
                        i4p = m4.laplace_info()
                        out["config4_vecchia_laplace"][pcname + "_preconditioner"] = {
                            "s_per_eval": s4p, "negll": v4p, "newton_it": i4p["newton_it"], "cg_it": i4p["cg_it"], "lanczos_it": i4p["lanczos_it"],
                            "ms_mode_finding": i4p["ms_mode"], "ms_logdet": i4p["ms_logdet"],
                            "note": "negll differs from the vadu value by the stochastic part of the log-determinant only (another preconditioner = other probe vectors, likelihoods.h:16407-16437)"}
                    except Exception as e:
                        out["config4_vecchia_laplace"][pcname + "_preconditioner"] = {"error": "%s: %s" % (type(e).__name__, e)}
                del m4
                out["config4_vecchia_laplace"]["roofline_kernels_from_trace"] = laplace_kernel_rooflines(n4, 30)
                for kk in ("roofline_cg_iteration", "roofline_logdet_iteration"):
                    out["config4_vecchia_laplace"][kk]["time_source"] = "host timer of the phase / iteration count (all kernels of an iteration + launch gaps); per-kernel figures from kernel durations: roofline_kernels_from_trace"
            except Exception as e:
                out["config4_vecchia_laplace"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_extras:
            # full-scale Vecchia / VIF: one likelihood evaluation at n = 1e5, m = 30, 200 inducing points
            try:
                nv = 100000
                rngv = np.random.default_rng(1)
                cv = rngv.uniform(size=(nv, 2)); yv = rngv.standard_normal(nv)
                tv0 = time.perf_counter()
                mv = gpboost_amd.GPModel(gp_coords=cv, cov_function="exponential", gp_approx="full_scale_vecchia", num_neighbors=30, num_ind_points=200,
                                         vecchia_ordering="random", seed=1)
                tv_setup = time.perf_counter() - tv0
                mv.neg_log_likelihood(np.array([0.1, 1.0, 0.1]), yv)
                tv1 = time.perf_counter()
                for kv in range(3):
                    vv = mv.neg_log_likelihood(np.array([0.1, 1.0 + 0.01 * (kv + 1), 0.1]))
                sv = (time.perf_counter() - tv1) / 3
                mv.neg_log_likelihood_and_gradient(np.array([0.1, 1.0, 0.1]), yv)
                tv2 = time.perf_counter()
                for kv in range(3):
                    mv.neg_log_likelihood_and_gradient(np.array([0.1, 1.0 + 0.01 * (kv + 1), 0.1]), yv)
                sgv = (time.perf_counter() - tv2) / 3
                out["vif_full_scale_vecchia"] = {
                    "workload": "Gaussian nll, gp_approx=full_scale_vecchia, n=%d, d=2, exponential, m=30, 200 inducing points (kmeans++ start on the host, Lloyd iterations on the device)" % nv,
                    "s_per_eval": sv, "s_per_eval_with_analytic_gradient": sgv, "negll": vv, "setup_s_incl_kmeans_and_device_neighbor_search": round(tv_setup, 3)}
                del mv
            except Exception as e:
                out["vif_full_scale_vecchia"] = {"error": "%s: %s" % (type(e).__name__, e)}
            # round 6: full-scale Vecchia with a NON-GAUSSIAN likelihood (FindModePostRandEffCalcMLLFSVA, likelihoods.h:3379-3750): Bernoulli-logit at
            # config 4's size with 200 inducing points, iterative methods with the reference's default preconditioner for these models ("fitc", 200 inducing points of its own):
            # one evaluation (mode finding + stochastic log-determinant, 50 probes) and one evaluation with the gradient wrt the covariance parameters
            try:
                nq = 100000
                rngq = np.random.default_rng(1)
                cq = rngq.uniform(size=(nq, 2))
                latq = 0.9 * np.sin(5 * cq[:, 0]) * np.cos(3 * cq[:, 1])
                yq = (rngq.uniform(size=nq) < 1.0 / (1.0 + np.exp(-1.5 * latq))).astype(np.float64)
                tq0 = time.perf_counter()
                mq = gpboost_amd.GPModel(likelihood="bernoulli_logit", gp_coords=cq, cov_function="exponential", gp_approx="full_scale_vecchia", num_neighbors=30,
                                         num_ind_points=200, vecchia_ordering="random", seed=1)
                tq_setup = time.perf_counter() - tq0
                mq.neg_log_likelihood(np.array([1.0, 0.1]), yq)           # level schedules, probe vectors, the preconditioner's inducing points
                tq1 = time.perf_counter()
                vq = mq.neg_log_likelihood(np.array([1.01, 0.1]), yq)
                sq = time.perf_counter() - tq1
                iq = mq.laplace_info()
                tq2 = time.perf_counter()
                mq.fit(yq, params={"optimizer_cov": "lbfgs", "init_cov_pars": np.array([1.0, 0.1]), "maxit": 2})
                sfit = time.perf_counter() - tq2
                out["vif_non_gaussian"] = {
                    "workload": "Bernoulli-logit nll, gp_approx=full_scale_vecchia, n=%d, d=2, exponential, m=30, 200 inducing points, fitc preconditioner (200 inducing points), 50 probes" % nq,
                    "s_per_eval": sq, "negll": vq, "newton_it": iq["newton_it"], "cg_it": iq["cg_it"], "lanczos_it": iq["lanczos_it"], "ms_factor": iq.get("ms_factor"),
                    "ms_mode_finding": iq["ms_mode"], "ms_logdet": iq["ms_logdet"], "setup_s": round(tq_setup, 3),
                    "s_two_lbfgs_iterations_with_gradients": sfit, "num_it": mq.get_num_optim_iter(), "cov_pars_after_two_iterations": [float(x) for x in mq.get_cov_pars()],
                    "roofline_kernels_from_trace": vif_kernel_rooflines(nq, 30, 200)}
                del mq
            except Exception as e:
                out["vif_non_gaussian"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_extras:
            # the direct caller of the hot path: a complete maximum-likelihood fit of (sigma2, sigma1_2, rho) on the
            # bench model -- y uploaded once, 3 / 7 doubles back per evaluation.  y = smooth signal + noise so that the optimum is interior.
            try:
                yf = np.sin(4 * coords[:, 0]) + 0.5 * rng.standard_normal(n)
                st.set_shard(0, n)
                tf = time.perf_counter()
                mdl.fit(yf, params={"optimizer_cov": "lbfgs"})
                sf = time.perf_counter() - tf
                oi = mdl.optim_info()
                out["fit_covariance_parameters"] = {
                    "workload": "GPB_OptimCovPar, lbfgs (reference default), n=%d, m=%d, %s: the bench model, y = sin(4 x0) + 0.5 eps" % (n, m, args.cov),
                    "s_per_fit": sf, "num_it": mdl.get_num_optim_iter(), "launches_likelihood_only": oi["num_ll_evals"],
                    "launches_with_gradient": oi["num_grad_evals"], "cov_pars": [float(v) for v in mdl.get_cov_pars()],
                    "negll": mdl.get_current_neg_log_likelihood()}
            except Exception as e:
                out["fit_covariance_parameters"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_extras:
            # BASELINE config 3: the device work of one GPBoost iteration (Gaussian likelihood, Vecchia m = 30, n = 1e5; tree of 31 leaves
            # on F = 50 features x 255 bins), every step through the C ABI with host arrays in and out (scripts/gpu_boost_iter.py)
            try:
                n3, F3, nb3, L3 = 100000, 50, 255, 31
                rng3 = np.random.default_rng(1)
                c3 = rng3.uniform(size=(n3, 2)); X3 = rng3.uniform(size=(n3, F3))
                y3 = np.sin(4 * X3[:, 0]) + X3[:, 1] ** 2 + 0.5 * rng3.standard_normal(n3)
                bins3 = np.minimum((X3 * (nb3 - 1)).astype(np.int64) + 1, nb3 - 1).astype(np.uint8).T.copy()
                gnb3 = np.full(F3, nb3, dtype=np.int32)
                bo3 = np.concatenate([[0], np.cumsum(gnb3)]).astype(np.int32)
                m3 = gpboost_amd.GPModel(gp_coords=c3, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="random", seed=1)
                cp3 = np.array([0.25, 0.1, 0.1])
                m3.set_optim_params({"optimizer_cov": "gradient_descent", "maxit": 1, "init_cov_pars": cp3})
                hb3 = shim.HistBuilder(bins3, bo3)
                hb3.pool_resize(L3 + 1)
                hb3.set_fix_info((bo3[:-1] + 1).astype(np.int32), np.full(F3, nb3, dtype=np.int32), np.zeros(F3, dtype=np.int32))
                hb3.set_split_info(np.ones(F3, dtype=np.int32), np.zeros(F3, dtype=np.int32), np.zeros(F3, dtype=np.int32))
                score = np.zeros(n3); t3 = {}
                R3 = 100                                  # BASELINE config 3 as written: 100 trees, timed as a whole
                t_all0 = time.perf_counter()
                for it3 in range(R3):
                    ta = time.perf_counter()
                    grad3 = m3.y_aux(m3.get_cov_pars() if it3 else cp3, score - y3)
                    tb = time.perf_counter()
                    hb3.set_gradients(grad3, None)           # upload of the n gradients (pageable numpy array: the caller may register its buffers) + max-abs pass
                    tb2 = time.perf_counter()
                    tree3 = hb3.grow_tree(L3, float(np.cumsum(grad3)[-1]), float(n3), 0.0, 20, 1e-3, 0.0)
                    tc = time.perf_counter()
                    vals3 = m3.newton_update_leaf_values(None, None, tree3["data_leaf_index"], tree3["num_leaves"])
                    td = time.perf_counter()
                    score = score + 0.1 * vals3[tree3["data_leaf_index"]]
                    te = time.perf_counter()
                    m3.fit(y3 - score)
                    tf3 = time.perf_counter()
                    t3 = {"gradient_yaux_ms": (tb - ta) * 1e3, "set_gradients_ms": (tb2 - tb) * 1e3, "tree_31_leaves_ms": (tc - tb2) * 1e3,
                          "newton_leaf_values_ms": (td - tc) * 1e3, "cov_par_step_ms": (tf3 - te) * 1e3}
                t_all = time.perf_counter() - t_all0
                t3["total_ms"] = sum(t3.values())
                out["config3_boosting_iteration"] = dict(
                    workload="GPBoost boosting loop as BASELINE config 3 writes it: %d trees, n=%d, F=%d, %d bins, %d leaves, Vecchia m=30 -- natively through this library's C ABI "
                             "(synthetic equal-width bins in the reference's layout); the *_ms entries are the LAST iteration's" % (R3, n3, F3, nb3, L3),
                    config3_100_trees_s=round(t_all, 4), ms_per_iteration_mean=round(t_all / R3 * 1e3, 3), **{k: round(v, 3) for k, v in t3.items()})
                hb3.close(); del m3
                # Route B, TIMED HERE (no quoted numbers): the same 100 iterations through the reference's own Booster / GBDT / REModel host code
                # (integration/_build/lib_gpboost_hip.so; integration/routeb_driver.py) with GPU_use = true, once with the reference's CPU tree learner and once with
                # device_type = gpu (whole trees on the device too), and ONE iteration of the same build's CPU path (GPU_use = false) beside them.
                try:
                    from integration import routeb_driver as rbd
                    if not rbd.available():
                        out["config3_boosting_iteration"]["route_b"] = {"skipped": "integration/_build/lib_gpboost_hip.so is not built on this box (make -C oracle routeB needs /root/reference)"}
                    else:
                        rb = rbd.RouteB()
                        y3b = y3 + np.sin(5 * c3[:, 0]) * np.cos(4 * c3[:, 1])
                        ra = rb.boosting_loop(c3, X3, y3b, R3, gpu_use=True, device_trees=False)
                        rt = rb.boosting_loop(c3, X3, y3b, R3, gpu_use=True, device_trees=True)
                        rc = rb.boosting_loop(c3, X3, y3b, 1, gpu_use=False, device_trees=False)
                        out["config3_boosting_iteration"]["route_b"] = {
                            "workload": "the same configuration through the reference's own Booster (LGBM_BoosterUpdateOneIter x %d, its own bin mappers), measured in this run" % R3,
                            "gpu_use_true_100_trees_s": round(ra["loop_s"], 4), "gpu_use_true_ms_per_iteration_median": round(1e3 * float(np.median(ra["per_iteration_s"])), 3),
                            "gpu_use_true_device_trees_100_trees_s": round(rt["loop_s"], 4), "gpu_use_true_device_trees_ms_per_iteration_median": round(1e3 * float(np.median(rt["per_iteration_s"])), 3),
                            "setup_s": round(ra["setup_s"], 3), "device_trees_vs_host_trees_max_abs_prediction_diff": float(np.abs(rt["pred"] - ra["pred"]).max()),
                            "cpu_path_same_build_s_first_iteration": round(rc["loop_s"], 3),
                            "cpu_path_note": "GPU_use = false, device_type = cpu of the same library: ONE boosting iteration (its first; the loop is not run to 100 on the CPU here)"}
                        # round 6: the same loop at n = 1e6 (50 features, 31 leaves, device trees, covariance parameters trained in the loop): 12 iterations, median of the last 10
                        try:
                            rng6 = np.random.default_rng(11)
                            n6 = 1000000
                            c6 = rng6.uniform(size=(n6, 2)); X6 = np.ascontiguousarray(rng6.uniform(size=(n6, 50)))
                            y6 = np.sin(4 * X6[:, 0]) + X6[:, 1] ** 2 + np.sin(5 * c6[:, 0]) * np.cos(4 * c6[:, 1]) + 0.3 * rng6.standard_normal(n6)
                            r6 = rb.boosting_loop(c6, X6, y6, 12, gpu_use=True, device_trees=True)
                            out["config3_boosting_iteration"]["route_b"]["n1e6_device_trees"] = {
                                "workload": "n=1000000, 50 features, 255 bins, 31 leaves, Vecchia m=30, train_gp_model_cov_pars=true: LGBM_BoosterUpdateOneIter x 12 through the reference's Booster",
                                "ms_per_iteration_median_of_last_10": round(1e3 * float(np.median(r6["per_iteration_s"][2:])), 3),
                                "ms_per_iteration_min_max_of_last_10": [round(1e3 * float(np.min(r6["per_iteration_s"][2:])), 3), round(1e3 * float(np.max(r6["per_iteration_s"][2:])), 3)],
                                "setup_s_model_dataset_booster": round(r6["setup_s"], 3)}
                            del X6, c6, y6, r6
                        except Exception as e:
                            out["config3_boosting_iteration"]["route_b"]["n1e6_device_trees"] = {"error": "%s: %s" % (type(e).__name__, e)}
                except Exception as e:
                    out["config3_boosting_iteration"]["route_b"] = {"error": "%s: %s" % (type(e).__name__, e)}
            except Exception as e:
                out["config3_boosting_iteration"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_extras:
            # shard_sweep (VERDICT r05 #1 iv / #11): each of the 8 shards an 8-rank job evaluates (parallel.shard_range), run ALONE on this one device -- point-kernel
            # time by HIP events and the wall time of one synchronous evaluation of the shard (launch + in-launch sums + pinned-memory hand-over).  A ONE-DEVICE
            # PROJECTION of the per-rank work of an N = 8 launch (later shards do the same arithmetic on a different gather footprint), never a scaling number.
            def sweep(state, cti, vv, aa, nn_pts, label):
                rows = []
                state.set_shard(0, nn_pts)
                state.bench(shim.MODE_NLL, cti, vv, aa, 0, 60)          # clocks at their loaded state before the first shard is timed (the first shard of a cold sweep read 15 % high)
                for r8 in range(8):
                    j0, j1 = parallel.shard_range(nn_pts, r8, 8)
                    state.set_shard(j0, j1)
                    _, kms, _ = state.bench(shim.MODE_NLL, cti, vv, aa, 3, 20)
                    _, gms, _ = state.bench(shim.MODE_GRAD, cti, vv, aa, 2, 10)
                    for _w in range(3):
                        state.nll_terms(cti, vv, aa)
                    tw = time.perf_counter()
                    for kk in range(20):
                        state.nll_terms(cti, vv * (1. + 1e-3 * kk), aa)
                    rows.append({"rank": r8, "points": j1 - j0, "nll_kernel_ms": round(kms, 5), "grad_kernel_ms": round(gms, 5),
                                 "nll_step_ms_synchronous": round((time.perf_counter() - tw) / 20 * 1e3, 5)})
                state.set_shard(0, nn_pts)
                _, kfull, _ = state.bench(shim.MODE_NLL, cti, vv, aa, 3, 20)
                ks = [r_["nll_kernel_ms"] for r_ in rows]; ss = [r_["nll_step_ms_synchronous"] for r_ in rows]
                return {"workload": label, "label": "ONE-DEVICE PROJECTION: every shard of an 8-rank job run alone on this GPU; not a scaling measurement",
                        "shards": rows, "one_launch_all_points_kernel_ms": round(kfull, 5), "slowest_shard_kernel_ms": max(ks), "slowest_shard_step_ms": max(ss),
                        "projected_speedup_8_ranks_kernel_only": round(kfull / max(ks), 3),
                        "projected_speedup_8_ranks_step": round((dt / args.steps * 1e3) / max(ss), 3) if (n, m, d) == (nn_pts, state.m, state.d) else None}
            try:
                out["shard_sweep"] = {"metric_shape": sweep(st, ct, var0, a0, n, "n=%d, d=%d, %s, m=%d" % (n, d, args.cov, m))}
                if (n, d, m, args.cov) == (1000000, 2, 30, "exponential"):
                    rng5 = np.random.default_rng(1)
                    c5 = rng5.uniform(size=(1000000, 3)); y5 = rng5.standard_normal(1000000)
                    m5 = gpboost_amd.GPModel(gp_coords=c5, cov_function="matern", cov_fct_shape=2.5, gp_approx="vecchia", num_neighbors=40, vecchia_ordering="random", seed=1)
                    for _w in range(3):
                        m5.neg_log_likelihood(cov_pars, y5 if _w == 0 else None)        # y resident; third evaluation: the spatially sorted gather copy
                    st5 = shim.VecchiaState.from_handle(m5.vecchia_handle(), 1000000, 3, 40)
                    out["shard_sweep"]["config5_shape"] = sweep(st5, 2, var0, np.sqrt(5.0) / cov_pars[2], 1000000, "BASELINE config 5: n=1e6, d=3, Matern-2.5, m=40")
                    del m5
            except Exception as e:
                out["shard_sweep"] = dict(out.get("shard_sweep", {}), error="%s: %s" % (type(e).__name__, e))
        if world == 1 and not args.no_cpu_baseline:
            # (1) the reference's CPU path on the host cores with the GPU IDLE (round 3 timed it beside a loop that drove the GPU flat out)
            try:
                out["cpu_baseline"] = cpu_baseline(coords, y, cov_function, shape, m, cov_pars, n)
            except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            # (2) a SUSTAINED rate over ~10 s with an idle host (clocks / thermals at steady state): the batched entry point in a loop; reported
            # next to the `steps`-step `value`, never instead of it
            try:
                cps32 = np.stack([cov_pars_of(k) for k in range(32)])
                t0s = time.perf_counter(); nev = 0
                while time.perf_counter() - t0s < 10.0:
                    mdl.neg_log_likelihood_batch(cps32); nev += 32
                dts = time.perf_counter() - t0s
                out["config"]["sustained"] = {"evals_per_s": nev / dts, "seconds": round(dts, 1), "evals": nev,
                                              "call": "GPB_HIP_EvalNegLogLikelihoodBatch (K = 32) in a loop for 10 s, host otherwise idle (after the CPU baseline leg)"}
            except Exception as e:   # noqa: BLE001
                out["config"]["sustained"] = {"error": "%s: %s" % (type(e).__name__, e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
