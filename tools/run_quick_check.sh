set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest40.log 2>&1 || { tail -30 gpurun_out/pytest40.log; exit 1; }
tail -2 gpurun_out/pytest40.log
timeout -k 10 300 python bench.py --cpu-iters 0 --no-extras > gpurun_out/bench40.json
timeout -k 10 300 python bench.py --cpu-iters 0 --no-extras --rows 1000000 --cols 16 --steps 2000 > gpurun_out/bench40_cfg2.json
python - <<'PY'
import json
for f in ("gpurun_out/bench40.json","gpurun_out/bench40_cfg2.json"):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"])
PY
