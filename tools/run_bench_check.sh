set -e
timeout -k 10 400 python bench.py > gpurun_out/bench_n100m.json 2> gpurun_out/bench_n100m.err || { tail -20 gpurun_out/bench_n100m.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open("gpurun_out/bench_n100m.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["achieved"], j["extra"]["n100m"])
PY
MHLIK_BENCH_BACKEND=gloo MHLIK_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 --big-rows 20000000 > gpurun_out/bench_rehearse2.json 2> gpurun_out/bench_rehearse2.err || { tail -20 gpurun_out/bench_rehearse2.err; exit 1; }
tail -1 gpurun_out/bench_rehearse2.json | cut -c1-1500
