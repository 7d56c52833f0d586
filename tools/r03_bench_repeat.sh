set -e
for i in 1 2 3; do
  timeout -k 10 100 python bench.py --cpu-iters 0 --no-extras --no-kchain > gpurun_out/rep300_$i.json
  timeout -k 10 100 python bench.py --cpu-iters 0 --no-extras --no-kchain --steps 20 --warmup 5 > gpurun_out/rep20_$i.json
done
timeout -k 10 100 python tools/chain_overhead_c.py 2>&1 | grep "K=300" | head -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/rep*_?.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["value"],1), round(j["ms_per_step"]*1e3,1), round(j["roofline"]["kernel_ms"]*1e3,1), round((j["ms_per_step"]-j["roofline"]["kernel_ms"])*1e3,1))
PY
