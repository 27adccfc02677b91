#!/bin/bash
# ONE parametrised runner for the GPU box (replaces the one-shot scripts/gpu_r0*.sh / profile_r0*.sh of rounds 1-4; their history is in git).
#   gpurun --timeout S -- 'bash scripts/gpu_run.sh <tag> <step> [<step> ...]'      -> gpurun_out/<tag>/...
# Steps (each bounded by its own timeout; a failing step does not stop the ones after it):
#   pytest:<pytest args>         python -m pytest <args> -m gpu -q              -> pytest_<k>.log
#   suite                        the whole -m gpu suite                         -> pytest_gpu.log
#   smoke                        __graft_entry__.smoke()                        -> smoke.log
#   bench:<name>:<bench args>    python bench.py <args>                         -> bench_<name>.json (+ .err)
#   rehearsal:<N>[,<N>...]       python bench.py --gpus N (self-launching; rehearses on one device)  -> bench_rehearsal_N<N>.json
#   dist1                        bench.py under torch.distributed.run with ONE rank and GPB_BENCH_FORCE_DIST=1 (RCCL + mailbox + A/B in the loop) -> bench_forced_dist_1rank.json
#   config5                      bench.py at config 5's shape (n=1e6, d=3, Matern-2.5, m=40), metric line only -> bench_config5.json
#   ubench:<name>                scripts/ubench/<name> (prebuilt binary)        -> ubench_<name>.log
#   py:<name>:<script + args>    python scripts/<script> <args>                 -> <name>.log
#   trace:<name>:<bench args>    rocprofv3 --kernel-trace --stats around bench.py <args>   -> <name>_rocprofv3_summary.txt, <name>_under_trace.json
#   pmc[:<target args>]          the PMC passes (each its own run, no trace domain) over scripts/gpu_pmc_target.py  -> pmc.json, pmc_summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
k=0
show() { grep -v "^$" "$1" | tail -${2:-8} | cut -c1-${3:-300}; }
for STEP in "$@"; do
  k=$((k+1))
  KIND=${STEP%%:*}; REST=${STEP#*:}; [ "$KIND" = "$STEP" ] && REST=""
  echo "=== [$k] $STEP"
  case $KIND in
    pytest)    (time timeout 1500 python -m pytest $REST -m gpu -q) > $O/pytest_$k.log 2>&1; show $O/pytest_$k.log 12 ;;
    suite)     (time timeout 2400 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1; show $O/pytest_gpu.log 12 ;;
    smoke)     (time timeout 600 python __graft_entry__.py smoke) > $O/smoke.log 2>&1; show $O/smoke.log 4 600 ;;
    bench)     NAME=${REST%%:*}; ARGS=${REST#*:}; [ "$NAME" = "$REST" ] && ARGS=""
               timeout 900 python bench.py $ARGS > $O/bench_$NAME.json 2> $O/bench_$NAME.err; echo "rc=$?"; cut -c1-1500 $O/bench_$NAME.json; show $O/bench_$NAME.err 4 ;;
    rehearsal) for N in ${REST//,/ }; do
                 timeout 900 python bench.py --gpus $N --steps 20 --warmup 5 > $O/bench_rehearsal_N$N.json 2> $O/bench_rehearsal_N$N.err; echo "N=$N rc=$?"
                 cut -c1-900 $O/bench_rehearsal_N$N.json; grep -v "^$" $O/bench_rehearsal_N$N.err | grep -iv "warn\|OMP_NUM\|\*\*\*" | tail -4 | cut -c1-300
               done ;;
    dist1)     # the multi-GPU code path with ONE rank under the driver's launcher: nccl process group, in-library RCCL communicator, mailbox, the A/B of the sums through ncclAllReduce
               GPB_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_forced_dist_1rank.json 2> $O/bench_forced_dist_1rank.err
               echo "rc=$?"; cut -c1-1500 $O/bench_forced_dist_1rank.json; grep -v "^$" $O/bench_forced_dist_1rank.err | grep -iv "warn\|OMP_NUM\|\*\*\*" | tail -4 | cut -c1-300 ;;
    config5)   timeout 900 python bench.py --n 1000000 --m 40 --d 3 --cov matern_2.5 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_config5.json 2> $O/bench_config5.err
               echo "rc=$?"; cut -c1-1800 $O/bench_config5.json; show $O/bench_config5.err 4 ;;
    ubench)    timeout 300 scripts/ubench/$REST > $O/ubench_$REST.log 2>&1; echo "rc=$?"; show $O/ubench_$REST.log 60 ;;
    py)        NAME=${REST%%:*}; ARGS=${REST#*:}
               (time timeout 1500 python scripts/$ARGS) > $O/$NAME.log 2>&1; echo "rc=$?"; show $O/$NAME.log 30 400 ;;
    trace)     NAME=${REST%%:*}; ARGS=${REST#*:}; [ "$NAME" = "$REST" ] && ARGS=""
               R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$NAME -- python $R/bench.py $ARGS > $R/$O/${NAME}_under_trace.json 2> $R/$O/trace_$NAME.err)
               python scripts/summarize_prof.py trace $O/trace_$NAME > $O/${NAME}_rocprofv3_summary.txt 2>&1; head -16 $O/${NAME}_rocprofv3_summary.txt | cut -c1-260; rm -rf $O/trace_$NAME ;;
    pmc)       R=$PWD; T="python $R/scripts/gpu_pmc_target.py $REST"; i=0
               for C in "FETCH_SIZE" "WRITE_SIZE" \
                        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
                        "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F64" \
                        "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
                 i=$((i+1))
                 (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc$i -- $T > $R/$O/pmc$i.log 2> $R/$O/pmc$i.err) || echo "pmc pass $i ($C) failed: $(tail -2 $O/pmc$i.err)"
               done
               python scripts/summarize_prof.py pmc $O/pmc[0-9] > $O/pmc_summary.txt
               python scripts/summarize_prof.py pmc-json $O/pmc.json $O/pmc[0-9]
               grep -i "vecchia_point\|hist_build\|hist_reduce\|syrk_mfma\|dense_cov" $O/pmc_summary.txt | cut -c1-260 | head -40
               rm -rf $O/pmc[0-9] ;;
    tracepy)   # rocprofv3 --kernel-trace --stats around a python script of scripts/ (round 6: the round-5 kernels that had no trace)  -> <name>_rocprofv3_summary.txt, <name>.log
               NAME=${REST%%:*}; ARGS=${REST#*:}
               R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$NAME -- python $R/scripts/$ARGS > $R/$O/$NAME.log 2> $R/$O/trace_$NAME.err)
               python scripts/summarize_prof.py trace $O/trace_$NAME > $O/${NAME}_rocprofv3_summary.txt 2>&1; show $O/$NAME.log 4 400; head -14 $O/${NAME}_rocprofv3_summary.txt | cut -c1-230; rm -rf $O/trace_$NAME ;;
    pmcpy)     # the PMC passes (each its own run, no trace domain) over a python script of scripts/  -> <name>_pmc_summary.txt
               NAME=${REST%%:*}; ARGS=${REST#*:}
               R=$PWD; i=0
               for C in "FETCH_SIZE" "WRITE_SIZE" \
                        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
                        "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"; do
                 i=$((i+1))
                 (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmcpy$i -- python $R/scripts/$ARGS > $R/$O/pmcpy$i.log 2> $R/$O/pmcpy$i.err) || echo "pmc pass $i ($C) failed: $(tail -2 $O/pmcpy$i.err)"
               done
               python scripts/summarize_prof.py pmc $O/pmcpy[0-9] > $O/${NAME}_pmc_summary.txt
               head -24 $O/${NAME}_pmc_summary.txt | cut -c1-260
               rm -rf $O/pmcpy[0-9] $O/pmcpy[0-9].log ;;
    env)       # env:<VAR=VALUE>: exported for the steps after it
               export "$REST"; echo "exported $REST" ;;
    *)         echo "unknown step $STEP" ;;
  esac
done
ls -la $O | tail -30
