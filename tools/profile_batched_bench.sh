#!/bin/bash
# rocprofv3 kernel-time summaries of bench.py's K-chain workloads (the commands profiles/r02_batched_bench_lines.jsonl
# times), one run per workload: gpurun_out/<tag>_<workload>_kernel_stats.csv + the bench line printed under the profiler.
# usage: bash tools/profile_batched_bench.sh <tag>      (on the GPU box, from the repo root)
set -e
tag=${1:-r02}
R=$PWD
out=$R/gpurun_out
export TMPDIR=/tmp
run() { # name, bench args...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}_kt -- python3 $R/bench.py "$@" > $out/${tag}_${name}_line.json 2> $out/${tag}_${name}.err
  find $out/${tag}_${name}_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_${name}_kernel_stats.csv
  rm -rf $out/${tag}_${name}_kt
  head -3 $out/${tag}_${name}_kernel_stats.csv
}
run batched_f64 --workload batched_f64 --steps 200
run batched_f32_split --workload batched_f32 --steps 300
run batched_f32_exact --workload batched_f32 --f32-mode exact --steps 300
run mp_batched_f64 --workload mp_batched --resamp-factor 64 --steps 300
