#!/bin/bash
# Profiling recipe for the round's headline kernel (run on the GPU box via gpurun); outputs under gpurun_out/.
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-r01}
mkdir -p $OUT
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $B > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o bench -- $B > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o bench -- $B > /dev/null 2> $OUT/pmc_sq2.err
find $OUT -name "*.csv" | head -30
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
