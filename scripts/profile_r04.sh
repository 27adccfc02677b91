#!/bin/bash
# Round-4 evidence run on the GPU box (via gpurun): PMC passes (each in its own run, no trace domains) over scripts/gpu_pmc_target.py
# (metric configuration, config-5 shape, histogram at n = 1e7, exact GP at n = 16384) and a kernel trace of the default bench.
# Usage: scripts/profile_r04.sh <tag>   -> gpurun_out/<tag>/{pmc.json,pmc_summary.txt,bench_trace_summary.txt,...}
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r04}
OUT=gpurun_out/$TAG; mkdir -p $OUT
T="python scripts/gpu_pmc_target.py"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F64" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -- $T > $OUT/pmc$i.log 2> $OUT/pmc$i.err || echo "pmc pass $i ($C) failed: $(tail -2 $OUT/pmc$i.err)"
done
python scripts/summarize_prof.py pmc $OUT/pmc* > $OUT/pmc_summary.txt
python scripts/summarize_prof.py pmc-json $OUT/pmc.json $OUT/pmc*
grep -i "vecchia_point\|hist_build\|hist_reduce\|syrk_mfma\|dense_cov" $OUT/pmc_summary.txt | cut -c1-260 | head -60
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace_bench.err
python scripts/summarize_prof.py trace $OUT/trace_bench > $OUT/bench_trace_summary.txt; head -14 $OUT/bench_trace_summary.txt | cut -c1-260
rm -rf $OUT/pmc[0-9] $OUT/trace_bench
