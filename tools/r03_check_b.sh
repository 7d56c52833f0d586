set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharded.py tests/test_gpu_bench_paths.py tests/test_gpu_kchain.py tests/test_gpu_chain.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03b_pytest.log 2>&1 || { tail -60 gpurun_out/r03b_pytest.log; exit 1; }
tail -3 gpurun_out/r03b_pytest.log
