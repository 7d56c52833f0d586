#!/bin/bash
# Evidence for bench.py's roofline object, taken on the command bench.py's default run times (same binary, same path):
#   1. rocprofv3 --kernel-trace --stats        -> <tag>_bench_kernel_stats.csv (mean duration of the streaming kernel)
#   2. rocprofv3 --pmc FETCH_SIZE              -> bytes fetched per launch   } separate passes, as the guide prescribes
#   3. rocprofv3 --pmc WRITE_SIZE              -> bytes written per launch   }
# and tools/summarize_bench_traffic.py folds 2+3 into <tag>_bench_pmc_hbm_traffic.json, which names the exact
# instantiation that was timed.  Copy the three outputs into profiles/.
# usage: bash tools/profile_bench.sh <tag>      (on the GPU box, from the repo root)
set -e
tag=${1:-r02}
R=$PWD
out=$R/gpurun_out
export TMPDIR=/tmp
args="--steps 40 --warmup 4 --cpu-iters 0 --no-extras"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_bench_kt -- python3 $R/bench.py $args > $out/${tag}_bench_line_under_rocprofv3.json 2> $out/${tag}_bench_kt.err
find $out/${tag}_bench_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_bench_kernel_stats.csv
find $out/${tag}_bench_kt -name "*kernel_trace.csv" -delete
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_bench_fetch -- python3 $R/bench.py $args > $out/${tag}_bench_fetch.json 2> $out/${tag}_bench_fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_bench_write -- python3 $R/bench.py $args > $out/${tag}_bench_write.json 2> $out/${tag}_bench_write.err
python3 $R/tools/summarize_bench_traffic.py $out/${tag}_bench_fetch $out/${tag}_bench_write $out/${tag}_bench_line_under_rocprofv3.json \
  "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) --output-format csv -- python3 bench.py $args" > $out/${tag}_bench_pmc_hbm_traffic.json
rm -rf $out/${tag}_bench_fetch $out/${tag}_bench_write
head -3 $out/${tag}_bench_kernel_stats.csv
cat $out/${tag}_bench_pmc_hbm_traffic.json
