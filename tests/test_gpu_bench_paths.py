"""bench.py's two drivers of the MH step -- the C sampler loop (device-resident
chain) and the Python fallback loop used when the in-library RCCL communicator
cannot be set up -- walk the same chain: same seed, same accept count."""
import argparse
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fallback_loop_walks_the_same_chain_as_the_c_sampler():
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from mcmc_amd import Likelihood, F64
    from mcmc_amd import samplers as S

    N, d, K, W = 200_000, 16, 150, 10
    a = argparse.Namespace(seed=3)
    rwsd = 2.38 / np.sqrt(d) * 0.02

    def barrier():
        torch.cuda.synchronize()

    with Likelihood(N, d, precision=F64, timing=True) as lk:
        lk.fill_synthetic(12345)
        _, k1, acc_c = bench.run_c_sampler(a, lk, S, np, N, d, K, W, rwsd, barrier)
        _, k2, acc_py = bench.run_host_loop(a, lk, S, np, torch, None, torch.device("cuda", 0), 1, d, K, W, rwsd, barrier)
    assert acc_c == acc_py and 0 < acc_c < K
    assert k1 > 0 and k2 > 0


def _bench_line(cmd, env):
    import json
    import subprocess

    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_single_process_line_has_the_contract_fields():
    env = dict(os.environ)
    j = _bench_line([sys.executable, "bench.py", "--rows", "400000", "--cols", "16", "--steps", "40", "--warmup", "5",
                     "--cpu-iters", "6", "--big-rows", "1200000"], env)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 40 and j["dtype"] == "f64" and j["vs_baseline"] is None
    r = j["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel_ms"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["parity"]["accept_trace_equal"] and c["parity"]["samples_equal"]
    assert c["parity"]["max_rel_loglik_diff"] <= 1e-10
    big = j["extra"]["n100m"]
    assert "error" not in big and big["rows"] == 1200000 and big["it_per_s"] > 0


def test_bench_two_rank_flow_on_one_gpu():
    """The N>1 flow of bench.py (torch.distributed.run launch, row shards, barrier, max over ranks, one JSON line from
    rank 0, the N=100M-style side run) rehearsed with two ranks on this one GPU.  RCCL refuses two ranks on one device,
    so with --allow-fallback the ranks agree on the fallback loop and combine the segment sums through
    torch.distributed (gloo); the chain is the single-process chain: same acceptance."""
    env = dict(os.environ)
    env.update(MHLIK_BENCH_DEVICE="0")
    common = ["--rows", "400000", "--cols", "16", "--steps", "40", "--warmup", "5", "--cpu-iters", "0"]
    two = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                       "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2",
                       "--allow-fallback", "--big-rows", "1200000"] + common, env)
    one = _bench_line([sys.executable, "bench.py", "--no-extras"] + common, dict(os.environ))
    assert two["n_gpus"] == 2 and two["config"]["rows_per_gpu"] == 200000
    assert "fallback host loop" in two["config"]["parallelism"] and two["rccl_ranks"] == 0
    assert two["acceptance"] == one["acceptance"]
    # 400k x 16 fits the workgroups' registers: the single-process run is the persistent chain kernel
    assert one["rccl_ranks"] == 0 and one["roofline"]["kernel"] == "chain_persist_kernel<double, 16, 5>"
    big = two["extra"]["n100m"]
    assert "error" not in big and big["rows_per_gpu"] == 600000 and big["in_library_rccl"] is False


def test_bench_eight_rank_flow_on_one_gpu():
    """The 8-GPU node's shape rehearsed on this one GPU: `python bench.py --gpus 8` launches eight ranks itself, each
    holds ONE of the eight row segments, all pinned to device 0 (MHLIK_BENCH_DEVICE=0).  RCCL refuses eight ranks on
    one device, so with --allow-fallback they agree on the torch.distributed (gloo) host loop; the chain and its
    acceptance are the single-process run's, every rank reports itself on stderr, and the N=100M-style side run shards
    the same way."""
    import json
    import subprocess

    env = dict(os.environ)
    env.update(MHLIK_BENCH_DEVICE="0")
    common = ["--rows", "400000", "--cols", "16", "--steps", "30", "--warmup", "5", "--cpu-iters", "0"]
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--allow-fallback", "--big-rows", "800000"] + common,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    eight = json.loads(lines[0])
    one = _bench_line([sys.executable, "bench.py", "--no-extras"] + common, dict(os.environ))
    assert eight["n_gpus"] == 8 and eight["config"]["rows_per_gpu"] == 50000 and eight["rccl_ranks"] == 0
    assert "fallback host loop" in eight["config"]["parallelism"]
    assert eight["acceptance"] == one["acceptance"]
    big = eight["extra"]["n100m"]
    assert "error" not in big and big["rows_per_gpu"] == 100000 and big["in_library_rccl"] is False
    for r in range(8):  # "bench.py: rank 3 LOCAL_RANK=3 device 0 pci ... mhlik_comm_ranks=0"
        assert "bench.py: rank %d LOCAL_RANK=%d device 0 " % (r, r) in out.stderr, out.stderr[-3000:]
    # the same eight ranks without the flag: non-zero, no number
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--no-extras"] + common, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=280)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
    assert "in-library RCCL communicator could not be built" in bad.stderr, bad.stderr[-3000:]


def test_default_line_carries_the_kchain_side_measurements(monkeypatch):
    """extra.kchain (BASELINE configs[4] beside the metric) is built by kchain_measure, the function the --workload
    runs print: three objects with chain-it/s, kernel ms and a roofline object each."""
    import argparse

    sys.path.insert(0, ROOT)
    import bench

    a = argparse.Namespace(seed=0, data_seed=12345, f32_mode="split", resamp_factor=16)
    for wl, kw, bound in (("batched_f64", {}, "mfma"), ("batched_f32", {"f32_mode": "split"}, "hbm"),
                          ("mp_batched", {"resamp_factor": 64}, "hbm")):
        ln = bench.kchain_measure(a, wl, 200_000, 128, 64, 20, 3, **kw)
        assert ln["metric"] == "chain_iterations_per_sec" and ln["value"] > 0 and ln["config"]["chains"] == 64
        r = ln["roofline"]
        assert r["bound"] == bound and r["kernel_ms"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


def test_bench_gpus_2_launches_itself_and_fails_loudly_without_rccl():
    """`python bench.py --gpus 2`, the way the driver calls the 1-GPU run: bench.py starts torch.distributed.run
    itself as a child process.  On a one-GPU box both ranks share device 0, RCCL refuses that
    (ncclCommInitRank: invalid usage), and without --allow-fallback the run must END NON-ZERO and print no JSON
    line -- never a headline number from the fallback loop."""
    import subprocess

    cmd = [sys.executable, "bench.py", "--gpus", "2", "--rows", "200000", "--cols", "16", "--steps", "10", "--warmup",
           "2", "--cpu-iters", "0", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=280)
    assert out.returncode != 0, out.stdout + out.stderr[-2000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout
    assert "in-library RCCL communicator could not be built" in out.stderr, out.stderr[-3000:]
    # and the same self-launched command with the explicit flag does report, labelled as the fallback
    j = _bench_line(cmd + ["--allow-fallback"], dict(os.environ))
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 0 and "fallback host loop" in j["config"]["parallelism"]


@pytest.mark.parametrize("workload", ["batched_f64", "mp_batched"])
def test_kchain_workloads_line_and_two_rank_refusal(workload):
    """bench.py's K-chain workloads: one JSON line with the contract fields on one GPU; with --gpus 2 on a one-GPU box
    the sharded set-up goes as far as RCCL's refusal of two ranks on one device and ends non-zero without a number
    (these workloads have no torch.distributed fallback)."""
    import subprocess

    env = dict(os.environ)
    base = [sys.executable, "bench.py", "--workload", workload, "--rows", "300000", "--cols", "16", "--chains", "20"]
    j = _bench_line(base + ["--steps", "30"], env)
    assert j["n_gpus"] == 1 and j["metric"] == "chain_iterations_per_sec" and j["value"] > 0
    assert j["roofline"]["kernel_ms"] > 0 and j["config"]["parallelism"] == "single GPU"
    out = subprocess.run(base + ["--steps", "10", "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "could not be built on every rank" in out.stderr
