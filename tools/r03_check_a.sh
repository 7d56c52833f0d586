set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_chain.py tests/test_gpu_samplers.py tests/test_gpu_golden.py tests/test_gpu_sharded.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r03a_pytest.log 2>&1 || { tail -40 gpurun_out/r03a_pytest.log; exit 1; }
tail -2 gpurun_out/r03a_pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a_bench20.json
timeout -k 10 300 python bench.py --cpu-iters 0 --no-extras > gpurun_out/r03a_bench300.json
timeout -k 10 300 python bench.py --cpu-iters 0 --no-extras --steps 20 --warmup 5 > gpurun_out/r03a_bench20b.json
python - <<'PY'
import json
for f in ("gpurun_out/r03a_bench20.json","gpurun_out/r03a_bench300.json","gpurun_out/r03a_bench20b.json"):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j["ms_per_step"]-j["roofline"]["kernel_ms"])*1e3, "us", j.get("extra",{}).get("sp"))
PY
