# Round check on the GPU box: full GPU test suite, smoke, the default bench line with its rocprofv3 kernel summary,
# the batched workloads, and the counter passes over every kernel.  Outputs under gpurun_out/.
set -e
tag=${1:-r01c}
out=gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1 || { tail $out/${tag}_smoke.log; exit 1; }
tail -1 $out/${tag}_smoke.log
timeout -k 10 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$out/${tag}_benchprof -- python3 $PWD/bench.py --no-extras --cpu-iters 0 > $out/${tag}_benchprof.json 2> $out/${tag}_benchprof.err
find $out/${tag}_benchprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_bench_kernel_stats.csv
find $out/${tag}_benchprof -name "*kernel_trace.csv" -delete
timeout -k 10 200 python bench.py --workload batched_f32 --steps 300 > $out/${tag}_bench_bf32_split.json
timeout -k 10 200 python bench.py --workload batched_f32 --f32-mode exact --steps 300 > $out/${tag}_bench_bf32_exact.json
timeout -k 10 200 python bench.py --workload batched_f64 --steps 100 > $out/${tag}_bench_bf64.json
timeout -k 10 200 python bench.py --workload mp_batched --resamp-factor 64 --steps 500 > $out/${tag}_bench_mpb64.json
timeout -k 10 200 python bench.py --workload mp_batched --steps 500 > $out/${tag}_bench_mpb16.json
bash tools/profile_kernels.sh ${tag}k > /dev/null
find $out -name "*kernel_trace.csv" -delete
cat $out/${tag}_bench.json
cat $out/${tag}_bench_bf32_split.json $out/${tag}_bench_bf32_exact.json $out/${tag}_bench_bf64.json $out/${tag}_bench_mpb64.json $out/${tag}_bench_mpb16.json | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['config']['workload'], j['config'].get('f32_mode'), j['config'].get('resamp_factor'), 'value %.0f  ms/step %.3f  kernel_ms %.3f acc %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['acceptance']))
"
head -4 $out/${tag}_bench_kernel_stats.csv
