# rocprofv3 kernel summaries of the persistent chain kernel at BASELINE configs[0] / [1] sizes (on the GPU box, repo root)
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out
for shape in "10000 3" "1000000 16"; do
  set -- $shape
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_persist_$1 -- python3 $PWD/bench.py --cpu-iters 0 --no-extras --rows $1 --cols $2 --steps 8000 --warmup 6000 > $out/r03_persist_$1.json 2> $out/r03_persist_$1.err
  find $out/r03_persist_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/r03_persist_$1_$2_kernel_stats.csv
  find $out/r03_persist_$1 -name "*kernel_trace.csv" -delete
  head -4 $out/r03_persist_$1_$2_kernel_stats.csv
  cat $out/r03_persist_$1.json
done
