#!/bin/bash
# Kernel-time and counter passes over every likelihood kernel at BASELINE sizes (tools/run_kernels.py).
# One rocprofv3 run per pass (counters are not split over passes); outputs under gpurun_out/<tag>_*.
# usage: bash tools/profile_kernels.sh <tag>
set -e
tag=${1:-prof}
R=$PWD
out=$R/gpurun_out
export TMPDIR=/tmp
run() { # name, rocprofv3 options...
  local name=$1; shift
  timeout -s KILL 170 rocprofv3 "$@" --output-format csv -d $out/${tag}_$name -- python3 $R/tools/run_kernels.py 4 > $out/${tag}_$name.log 2>&1
}
run kt --kernel-trace --stats
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run stalls --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
for n in mfma stalls fetch write; do python3 $R/tools/summarize_pmc.py $out/${tag}_$n $out/${tag}_$n.json "rocprofv3 --pmc (pass $n) -- python3 tools/run_kernels.py 4" > /dev/null; done
find $out/${tag}_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
head -30 $out/${tag}_kernel_stats.csv
