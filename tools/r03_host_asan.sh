#!/bin/bash
# libmhlik.so's HOST code under AddressSanitizer + UBSan on the GPU box (device code cannot be sanitized on this pool):
# the C drivers (no Python in the sanitized process: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and fails
# inside a Python process) walk create / upload / tuner / burn-in / device-resident chain (submit + collect pipeline,
# persistent kernel at this size) / statistics / destroy in all four precisions; tools/kchain_probe the K-chain samplers
# (double, float, mixed precision) and tools/chain_probe the chain C-ABI.
# -fno-sanitize=function: with the function sanitizer a kernel launched through a function pointer
# (auto kern = kernel<...>; hipLaunchKernelGGL(kern, ...)) is silently NOT launched by this toolchain.
# tools/ab/asan/libmhlik.so is built beforehand on the CPU:
#   hipcc --offload-arch=gfx950 -O3 -g -std=c++17 -fPIC -ffp-contract=off -pthread -Iinclude -Imcmc_amd/csrc \
#         -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=function -Xarch_host -fno-sanitize-recover=undefined \
#         -shared -o tools/ab/asan/libmhlik.so mcmc_amd/csrc/mhlik.hip -ldl
# The copy of the tree on the box is scratch: the product library is swapped in place there.
set -e
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
W=$(mktemp -d)
python3 tools/synthetic_csv.py -sz 5000 -dim 3 -seed 1 -o $W/small.csv   # with the product library, before the swap
python3 tools/synthetic_csv.py -sz 300000 -dim 8 -seed 2 -o $W/large.csv
cp tools/ab/asan/libmhlik.so mcmc_amd/lib/libmhlik.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:halt_on_error=1:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
run() { # binary, csv, rows, cols, tune, extra environment
  echo "== $1 $2 -tune $5 $6"
  env $6 LD_PRELOAD="${LD_PRELOAD:+$LD_PRELOAD:}$RT" GSL_RNG_SEED=7 timeout -k 10 300 mcmc_amd/bin/$1 -d 1 -sz $3 -dim $4 -samp 3000 \
    -burn 700 -lag 50 -tune $5 -in $W/$2.csv -out $W/${1}_ > $W/out 2> $W/err || { echo "FAILED rc=$?"; tail -40 $W/err; exit 1; }
  grep -v "amdgpu.ids" $W/err || true
  grep -E "Acceptance Ratio" $W/out | head -1
}
for b in mcmc_gpu mcmc_cpu mcmc_sp mcmc_mp; do
  run $b small 5000 3 0 MHLIK_CHAIN_PERSIST=1
  run $b small 5000 3 1 MHLIK_CHAIN_PERSIST=0
  run $b large 300000 8 0 MCMC_AMD_HOST_LOOP=0
  run $b large 300000 8 0 MCMC_AMD_HOST_LOOP=1
done
for e in MCMC_AMD_HOST_LOOP=0 MCMC_AMD_HOST_LOOP=1; do
  echo "== kchain_probe $e"
  env $e LD_PRELOAD="${LD_PRELOAD:+$LD_PRELOAD:}$RT" timeout -k 10 300 tools/kchain_probe 40000 6 16 > $W/kc.$e 2> $W/err || { echo "FAILED rc=$?"; tail -40 $W/err; exit 1; }
  grep -v "amdgpu.ids" $W/err || true
  cat $W/kc.$e
done
cmp $W/kc.MCMC_AMD_HOST_LOOP=0 $W/kc.MCMC_AMD_HOST_LOOP=1 || { echo "FAILED: device loop and host loop differ"; exit 1; }
echo "== chain_probe (C-ABI, one launch / two kernels)"
for e in MHLIK_CHAIN_PERSIST=1 MHLIK_CHAIN_PERSIST=0; do
  env $e LD_PRELOAD="${LD_PRELOAD:+$LD_PRELOAD:}$RT" timeout -k 10 120 tools/chain_probe 20000 5 64 2> $W/err | grep -v kernel_ms > $W/cp.$e || { echo "FAILED rc=$?"; tail -40 $W/err; exit 1; }
done
cmp $W/cp.MHLIK_CHAIN_PERSIST=1 $W/cp.MHLIK_CHAIN_PERSIST=0 || { echo "FAILED: one launch and two kernels differ"; exit 1; }
tail -2 $W/cp.MHLIK_CHAIN_PERSIST=1
echo "host sanitizers: clean"
