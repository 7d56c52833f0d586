# MP K-chain checks on the GPU box: parity tests, then the timing table (tools/bench_mp.py batched)
set -e
timeout -k 10 900 python -u -m pytest tests/test_gpu_mp_batched.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_mp.log 2>&1 || { tail -40 gpurun_out/pytest_mp.log; exit 1; }
tail -2 gpurun_out/pytest_mp.log
timeout -k 10 200 python tools/bench_mp.py batched > gpurun_out/mpb_dense.log 2>&1
cut -c1-150 gpurun_out/mpb_dense.log
