set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_mp_batched.py tests/test_gpu_batched.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest35.log 2>&1 || { tail -30 gpurun_out/pytest35.log; exit 1; }
tail -3 gpurun_out/pytest35.log
for m in exact split; do
  timeout -k 10 200 python bench.py --workload batched_f32 --f32-mode $m --steps 100 > gpurun_out/bench_bf32_$m.json
  MHLIK_BATCHED_F32=$m timeout -k 10 200 python bench.py --workload mp_batched --steps 100 > gpurun_out/bench_mpb_$m.json
  MHLIK_BATCHED_F32=$m timeout -k 10 200 python bench.py --workload mp_batched --resamp-factor 64 --steps 100 > gpurun_out/bench_mpb64_$m.json
done
cat gpurun_out/bench_bf32_*.json gpurun_out/bench_mpb*_*.json
