set -e
timeout -k 10 900 python -u -m pytest tests/test_gpu_batched.py tests/test_gpu_mp_batched.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest37.log 2>&1 || { tail -30 gpurun_out/pytest37.log; exit 1; }
tail -3 gpurun_out/pytest37.log
timeout -k 10 200 python bench.py --workload batched_f32 --steps 100 > gpurun_out/bench_bf32_split.json
timeout -k 10 200 python bench.py --workload mp_batched --resamp-factor 64 --steps 100 > gpurun_out/bench_mpb64_split.json
timeout -k 10 200 python bench.py --workload mp_batched --steps 100 > gpurun_out/bench_mpb_split.json
cat gpurun_out/bench_bf32_split.json gpurun_out/bench_mpb64_split.json gpurun_out/bench_mpb_split.json | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['config']['workload'], j['config'].get('f32_mode'), j['config'].get('resamp_factor'), 'value %.0f  ms/step %.3f  kernel_ms %.3f acc %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['acceptance']))
"
