#!/usr/bin/env python3
"""bench.py -- MH iterations/s and likelihood GB/s (% of HBM peak) at
N=10M, d=64 (BASELINE.json), on 1/2/4/8 MI355X.

A "step" is one Metropolis-Hastings iteration of the C sampler loop
(metropolis_gpu: d ziggurat draws, one double-precision likelihood pass over
the HBM-resident design matrix, prior, one uniform, accept test).  With
--gpus G > 1 the SAME N rows are sharded over the G ranks (strong scaling) and
every iteration's per-segment partial sums are all-reduced over RCCL.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus G --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--data-seed", type=int, default=12345)
    ap.add_argument("--cpu-iters", type=int, default=60, help="MH iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the SP / multi-core / N=100M side measurements")
    ap.add_argument("--big-rows", type=int, default=None,
                    help="rows of the BASELINE configs[3] side measurement (default: 100M beside the N=10M d=64 "
                         "metric, none for other sizes; 0 = skip)")
    ap.add_argument("--workload", default="mh", choices=["mh", "batched_f32", "batched_f64", "mp_batched"],
                    help="mh (default): BASELINE's metric, the DP MH step at N=10M d=64.  The others time BASELINE "
                         "configs[4] (64 lock-step chains, X.Theta on MFMA, N=10M d=128) on one GPU")
    ap.add_argument("--chains", type=int, default=64)
    ap.add_argument("--f32-mode", default="split", choices=["exact", "split"],
                    help="batched_f32: f32 MFMA (the single-theta kernel's fmaf chain) or three-bf16-piece split (default)")
    ap.add_argument("--resamp-factor", type=int, default=16, help="mp_batched: 1/F of the dark blocks re-examined per step")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="--gpus N>1: if the in-library RCCL communicator cannot be built, time the Python host loop that "
                         "all-reduces through torch.distributed instead of exiting non-zero (rehearsals on a one-GPU box)")
    ap.add_argument("--no-kchain", action="store_true", help="skip the BASELINE configs[4] side measurements (extra.kchain)")
    ap.add_argument("--kchain-steps", type=int, default=200, help="lock-step iterations of each extra.kchain measurement")
    ap.add_argument("--contested-iters", type=int, default=60,
                    help="MH iterations of the contested-decision parity check of the cpu_baseline leg (0 = skip)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start
    torch.distributed.run as a CHILD process (this process has not touched the
    GPU, and never will), relay its output and leave with its exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(child.stdout)
    sys.stdout.flush()
    raise SystemExit(child.returncode)


MFMA_PEAK_TFLOPS = {"batched_f32": 157.3, "batched_f64": 78.6, "mp_batched": 157.3}  # dense, MI355X_MICROARCH.md


def kchain_measure(a, workload, N, d, K, steps, W, rank=0, local_rank=0, world=1, dist=None, f32_mode=None,
                   resamp_factor=None):
    """One K-chain workload of BASELINE configs[4] through the C multichain samplers (whole lock-step iteration on the
    device): returns the JSON line as a dict (on every rank; rank 0 prints it)."""
    import numpy as np
    import torch

    from mcmc_amd import Likelihood
    from mcmc_amd import samplers as S

    f32_mode = f32_mode or a.f32_mode
    resamp_factor = resamp_factor or a.resamp_factor
    rwsd = 2.38 / np.sqrt(d) * 2e-3
    seeds = list(range(a.seed, a.seed + K))
    prec = {"batched_f32": 2, "batched_f64": 1, "mp_batched": 3}[workload]
    # mixed precision: one segment on one GPU (no slot pass); sharded, the fixed segments must hold whole 1024-row tiles
    mp_sharded = workload == "mp_batched" and world > 1
    with Likelihood(N, d, precision=prec, segments=1 if workload == "mp_batched" and world == 1 else 0, timing=True,
                    device=local_rank, rank=rank, world=world, segment_align=1024 if mp_sharded else 0) as lk:
        lk.fill_synthetic(a.data_seed)
        if world > 1:
            failed, why = 0, ""
            try:
                idt = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(Likelihood.make_comm_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                lk.comm_init(bytes(idt.numpy().tobytes()))
                if lk.comm_ranks() != world:
                    raise RuntimeError("communicator has %d ranks, expected %d" % (lk.comm_ranks(), world))
            except Exception as e:  # noqa: BLE001 -- reported on every rank
                why = str(e)
                print("rank %d: in-library RCCL set-up failed (%s)" % (rank, e), file=sys.stderr, flush=True)
                failed = 1
            t = torch.tensor([failed], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if int(t.item()) != 0:
                lk.comm_destroy()
                dist.barrier()
                dist.destroy_process_group()
                raise SystemExit("bench.py: rank %d: the in-library RCCL communicator could not be built on every rank "
                                 "(%s); no number is reported." % (rank, why or "another rank failed"))
            rank_report(rank, local_rank, lk.comm_ranks())
        if workload == "batched_f32":
            lk.set_batched_f32_mode(f32_mode)

        def run(n, burn):
            if workload == "mp_batched":
                return S.mp_multichain_sampler(lk, seeds, n, burn, rwsd, 64, N // 64, resamp_factor, 1.7e-7,
                                               shared_phase=True, theta0=np.zeros(d))
            return S.multichain_sampler(lk, seeds, n, burn, rwsd, use_f32=workload == "batched_f32")

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        run(W, 2)  # warm-up (mp: the all-bright first evaluations happen here)
        barrier()
        t0 = time.perf_counter()
        out = run(steps, 2)  # chain start + one burn-in iteration + `steps` sampling iterations
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        evals = steps + 2
        kernel_ms, _ = lk.last_kernel()  # device time of the last step's likelihood kernels (HIP events)
        rccl_ranks = lk.comm_ranks()
        local_rows = lk.info()["local_rows"]
    shard = local_rows / N if N else 1.0  # this rank's share of the rows: the kernel's bytes and flops per launch
    flops = 2.0 * N * d * K * shard
    achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[workload]
    roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": None, "kernel_ms": kernel_ms,
            "kernel": {"batched_f32": "lik_batched_f32_kernel", "batched_f64": "lik_batched_f64_kernel",
                       "mp_batched": "mpb_dark_kernel + mpb_fix_kernel + mpb_final_kernel (flops of the float pass)"}[workload],
            "algorithmic_flops_per_launch": flops}
    if workload == "mp_batched":
        # the float pass runs on the bf16 pipe (three-piece split) at 3/8 of the f32 MFMA cost, so the evaluation is
        # priced against HBM: the float matrix once, plus the double rows of the re-examined phase (1/F of the blocks
        # with one shared phase; the few bright blocks are not counted)
        nbytes = (N * d * 4 + N * d * 8 / resamp_factor) * shard
        gbps = nbytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "traffic": None, "kernel_ms": kernel_ms,
                "kernel": "mpb_dark_x3_kernel + mpb_fix_dense_kernel + mpb_final_kernel (one evaluation)",
                "algorithmic_bytes_per_launch": nbytes, "f32_equivalent_TFLOPs": achieved,
                "bf16_mfma_TFLOPs": 6 * achieved, "bf16_mfma_frac_of_dense_peak": 6 * achieved / 2500.0}
    if workload == "batched_f32" and f32_mode == "split":
        # six bf16 MFMAs per float product cost 3/8 of one f32 MFMA: the pass is no longer bound by the matrix pipe
        # but by the matrix read, so it is priced against HBM (the f32-equivalent rate is reported beside it)
        gbps = N * d * 4 * shard / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "traffic": None, "kernel_ms": kernel_ms, "kernel": "lik_batched_x3_kernel",
                "algorithmic_bytes_per_launch": N * d * 4 * shard, "f32_equivalent_TFLOPs": achieved,
                "bf16_mfma_TFLOPs": 6 * achieved, "bf16_mfma_frac_of_dense_peak": 6 * achieved / 2500.0}
    return {
        "metric": "chain_iterations_per_sec", "value": K * evals / elapsed, "unit": "chain-it/s", "n_gpus": world,
        "steps": steps, "warmup": W, "ms_per_step": elapsed / evals * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": {"batched_f32": "f32", "batched_f64": "f64", "mp_batched": "f32+f64"}[workload],
        "data": "synthetic",
        "config": {"workload": "%d lock-step MH chains, N=%d d=%d, %s (BASELINE configs[4])" % (K, N, d, workload),
                   "rows": N, "cols": d, "chains": K, "f32_mode": f32_mode if workload == "batched_f32" else None,
                   "resamp_factor": resamp_factor if workload == "mp_batched" else None,
                   "rows_per_gpu": local_rows,
                   "parallelism": "rows sharded over %d GPU(s), in-stream RCCL all-reduce of the K x segments slots" % world
                   if world > 1 else "single GPU"},
        "rccl_ranks": rccl_ranks,
        "roofline": roof,
        "acceptance": float(np.mean(out["accepted"])) / steps,
    }


INIT_TIMEOUT_S = 120  # rendezvous and every control-plane collective: a rank that died must not hang the others


def rank_report(rank, local_rank, comm_ranks):
    """One line per rank on stderr: enough to tell, from the log of a failed multi-GPU run, which process sat on
    which device and whether its communicator came up."""
    import torch

    try:
        dev = torch.cuda.current_device()
        p = torch.cuda.get_device_properties(dev)
        bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
        name = p.name
    except Exception as e:  # noqa: BLE001 -- a diagnostic must not end the run
        dev, bus, name = -1, "?", "(%s)" % e
    print("bench.py: rank %d LOCAL_RANK=%s device %d pci %s %s pid %d mhlik_comm_ranks=%d" %
          (rank, os.environ.get("LOCAL_RANK", "-"), dev, bus, name, os.getpid(), comm_ranks), file=sys.stderr, flush=True)


def main_batched(a):
    """BASELINE configs[4]: one step = one lock-step iteration of K chains (host
    proposals, one batched likelihood pass, host accept tests), the C loop of
    samplers/multichain.c.  Roofline: the X.Theta pass on the matrix cores."""
    import datetime

    import torch

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)  # before anything touches the GPU; does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the likelihood path has no CPU fallback")

    # Several GPUs: the rows are sharded, every rank runs the same K chains (same seeds) and the K x segments slots of
    # a pass are all-reduced in the stream by the library's own RCCL communicator; torch.distributed (gloo) only ships
    # the communicator id and takes the maximum of the rank times.  There is no fallback for these workloads.
    dist = None
    if world > 1:
        import torch.distributed as dist

        ndev = torch.cuda.device_count()
        local_rank = int(os.environ.get("MHLIK_BENCH_DEVICE", local_rank)) % ndev
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("MHLIK_BENCH_BACKEND", "gloo"),
                                timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))

    N = a.rows
    d = a.cols if a.cols != 64 else 128  # the configuration's d unless --cols says otherwise
    line = kchain_measure(a, a.workload, N, d, a.chains, a.steps, max(a.warmup, 2), rank, local_rank, world, dist)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


def main():
    a = parse()
    if a.workload != "mh":
        return main_batched(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)  # before anything touches the GPU; does not return
    import datetime

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the likelihood path has no CPU fallback")
    # torch.distributed is the CONTROL plane only (ships the 128-byte RCCL id, barrier, max of the rank times): gloo by
    # default, so that the one RCCL communicator on each device is the library's own, the one the data path uses.
    # MHLIK_BENCH_BACKEND=nccl puts the control plane on RCCL too.  MHLIK_BENCH_DEVICE pins every rank to one device
    # (rehearsal on a one-GPU box: RCCL refuses two ranks on one GPU, so that needs --allow-fallback).
    backend = os.environ.get("MHLIK_BENCH_BACKEND", "gloo")
    if "MHLIK_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MHLIK_BENCH_DEVICE"])
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        print("bench.py: rank %d: only %d GPU(s) visible, sharing device %d (RCCL will refuse several ranks on one "
              "device)" % (rank, ndev, local_rank % ndev), file=sys.stderr, flush=True)
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the tensors of torch collectives live
    allow_fallback = a.allow_fallback or os.environ.get("MHLIK_BENCH_ALLOW_FALLBACK") == "1"

    from mcmc_amd import Likelihood, F64, F32
    from mcmc_amd import samplers as S

    if world > 1:
        tmo = datetime.timedelta(seconds=INIT_TIMEOUT_S)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    N, d = a.rows, a.cols

    def make_context(rows):
        """The sharded double-precision context of this rank with device-generated data, and whether the
        in-library RCCL communicator is up on EVERY rank.  If it is not: every rank tears down what it built, and
        the run ends non-zero -- unless --allow-fallback asked for the torch.distributed host loop."""
        ctx = Likelihood(rows, d, precision=F64, device=local_rank, rank=rank, world=world, timing=True)
        ctx.fill_synthetic(a.data_seed)
        ok = True
        if world > 1:
            # the in-library path: ncclAllReduce on the context's stream, inside the device-resident chain
            failed, why = 0, ""
            try:
                if os.environ.get("MHLIK_BENCH_FORCE_FALLBACK") == "1":
                    raise RuntimeError("fallback forced")
                idt = torch.zeros(128, dtype=torch.uint8, device=red_dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(Likelihood.make_comm_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                ctx.comm_init(bytes(idt.cpu().numpy().tobytes()))
                if ctx.comm_ranks() != world:
                    raise RuntimeError("communicator has %d ranks, expected %d" % (ctx.comm_ranks(), world))
            except Exception as e:  # noqa: BLE001 -- reported below, on every rank
                why = str(e)
                print("rank %d: in-library RCCL set-up failed (%s)" % (rank, e), file=sys.stderr, flush=True)
                failed = 1
            t = torch.tensor([failed], dtype=torch.int32, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ok = int(t.item()) == 0
            if not ok:
                ctx.comm_destroy()  # a rank that did get a communicator must not all-reduce alone
                if not allow_fallback:
                    ctx.close()
                    dist.barrier()
                    dist.destroy_process_group()
                    raise SystemExit("bench.py: rank %d: the in-library RCCL communicator could not be built on every "
                                     "rank (%s); no number is reported.  --allow-fallback times the torch.distributed "
                                     "host loop instead." % (rank, why or "another rank failed"))
        elif os.environ.get("MHLIK_BENCH_FORCE_FALLBACK") == "1":
            ok = False
        return ctx, ok

    def timed_chain(ctx, ok, rows, steps, warmup):
        """warmup burn-in iterations, then exactly `steps` timed MH iterations; max over ranks."""
        step_sd = 2.38 / np.sqrt(d) * 2e-3  # small random-walk step: the posterior at N=1e7 is narrow
        if ok:
            el, kms, acc = run_c_sampler(a, ctx, S, np, rows, d, steps, warmup, step_sd, barrier)
        else:
            el, kms, acc = run_host_loop(a, ctx, S, np, torch, dist, red_dev, world, d, steps, warmup, step_sd, barrier)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, kms, acc

    lk, in_library_rccl = make_context(N)
    if world > 1:
        rank_report(rank, local_rank, lk.comm_ranks())
    K, W = a.steps, a.warmup
    rwsd = 2.38 / np.sqrt(d) * 2e-3
    elapsed, kernel_ms, n_accepted = timed_chain(lk, in_library_rccl, N, K, W)

    info = lk.info()
    local_bytes = info["local_rows"] * d * 8
    achieved = local_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    kernel_name = lk.last_kernel_name()  # the instantiation the timed steps launched
    rccl_ranks = lk.comm_ranks()

    traffic, traffic_src = pmc_traffic(N, d, info["local_rows"], kernel_name)
    out = {
        "metric": "mh_iterations_per_sec",
        "value": K / elapsed,
        "unit": "it/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "Bayesian logistic regression MH, N=%d d=%d DP (%s)" % (N, d, workload_name(N, d)),
                   "rows": N, "cols": d, "rows_per_gpu": info["local_rows"],
                   "parallelism": ("rows sharded over %d GPU(s), per-iteration RCCL all-reduce%s" %
                                   (world, "" if in_library_rccl else " through torch.distributed (fallback host loop)"))
                   if world > 1 else "single GPU"},
        "rccl_ranks": rccl_ranks,
        "likelihood_GBps_whole_job": N * d * 8 / (elapsed / K) / 1e9,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": kernel_name, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": local_bytes},
        "acceptance": n_accepted / K,
    }

    # ---- side measurements on rank 0 of a single-GPU run ----------------------------------
    if world == 1 and rank == 0:
        if not a.no_extras:
            out["extra"] = extras(a, lk, S, np, N, d)
        if a.cpu_iters > 0:
            out["cpu_baseline"] = cpu_baseline(a, np, N, d, rwsd, lk, S)
    lk.close()

    # ---- BASELINE configs[3] beside the metric's size, at every GPU count: N=100M, d=64, the same rows sharded ---
    if a.big_rows is None:
        a.big_rows = 100_000_000 if (N, d) == (10_000_000, 64) else 0
    if not a.no_extras and a.big_rows > 0:
        big = {"rows": a.big_rows, "cols": d}
        try:
            ctx, ok = make_context(a.big_rows)  # 51.2 GB on one GPU, 6.4 GB per GPU on 8
            steps_big = max(20, min(K, 100))
            el, kms, _ = timed_chain(ctx, ok, a.big_rows, steps_big, 3)
            rows_local = ctx.info()["local_rows"]
            ctx.close()
            big.update({"it_per_s": steps_big / el, "ms_per_step": el / steps_big * 1e3, "steps": steps_big,
                        "kernel_ms": kms, "rows_per_gpu": rows_local,
                        "kernel_GBps": rows_local * d * 8 / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                        "likelihood_GBps_whole_job": a.big_rows * d * 8 / (el / steps_big) / 1e9,
                        "in_library_rccl": bool(ok)})
        except Exception as e:  # noqa: BLE001 -- a side measurement never costs the main line
            big["error"] = str(e)
        out.setdefault("extra", {})["n100m"] = big
    # ---- BASELINE configs[4] beside the metric (single GPU, the metric's size only): 64 lock-step chains, d=128 ----
    if world == 1 and not a.no_extras and not a.no_kchain and (N, d) == (10_000_000, 64):
        kc = {}
        for name, wl, kw in (("batched_f64", "batched_f64", {}), ("batched_f32_split", "batched_f32", {"f32_mode": "split"}),
                             ("mp_batched_F64_shared_phase", "mp_batched", {"resamp_factor": 64})):
            try:
                ln = kchain_measure(a, wl, N, 128, 64, a.kchain_steps, 3, **kw)
                kc[name] = {"chain_it_per_s": ln["value"], "ms_per_step": ln["ms_per_step"], "steps": ln["steps"],
                            "kernel_ms": ln["roofline"]["kernel_ms"], "roofline": ln["roofline"],
                            "acceptance": ln["acceptance"], "config": ln["config"]}
            except (Exception, SystemExit) as e:  # noqa: BLE001 -- a side measurement never costs the main line
                kc[name] = {"error": str(e)}
        out.setdefault("extra", {})["kchain"] = kc
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_c_sampler(a, lk, S, np, N, d, K, W, rwsd, barrier):
    """W burn-in iterations, then K timed iterations of metropolis_gpu (the C
    sampler loop, device-resident chain).  Returns (seconds, mean kernel ms, accepted)."""
    lib = S.load_samplers()
    P = C.c_void_p
    lib.burn_in_metropolis_gpu.argtypes = [P, P, S.mcmc_str, S.mcmc_tune_str, S.mcmc_v_str, S.mcmc_int_v, P]
    lib.metropolis_gpu.argtypes = [P, P, S.mcmc_str, S.mcmc_tune_str, S.mcmc_v_str, S.mcmc_int_v, P,
                                   C.POINTER(C.c_int), C.POINTER(S.out_str)]
    burn_len = max(W, 1)  # burn-in does 1 + (burnin-1) = W likelihood evaluations
    samples = np.zeros((K, d))
    burn = np.zeros((burn_len, d))
    mcin = S.mcmc_str(ddata=d, Nd=N, Ns=K, burnin=burn_len, impl=S.IMPL_GPU)
    mct = S.mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    mcdata = S.mcmc_v_str(samples=samples.ctypes.data, burn=burn.ctypes.data)
    prop, cur = np.zeros(d), np.zeros(d)
    mclocv = S.mcmc_int_v(proposed=prop.ctypes.data, current=cur.ctypes.data)
    mcloc = S.mcmc_int()
    res = S.out_str()
    accepted = C.c_int(0)
    rng = S.Rng(a.seed)  # same seed on every rank: identical proposals and decisions everywhere
    lib.burn_in_metropolis_gpu(lk.handle, rng.r, mcin, mct, mcdata, mclocv, C.byref(mcloc))
    barrier()
    t0 = time.perf_counter()
    lib.metropolis_gpu(lk.handle, rng.r, mcin, mct, mcdata, mclocv, C.byref(mcloc), C.byref(accepted), C.byref(res))
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, float(res.kernelTime), accepted.value


def run_host_loop(a, lk, S, np, torch, dist, dev, world, d, K, W, rwsd, barrier):
    """Fallback when the in-library RCCL communicator cannot be set up: the same
    MH iteration (d ziggurat draws, likelihood, prior N(0,10^2), uniform, accept
    iff log u <= ratio) driven from Python, the per-segment sums all-reduced by
    torch.distributed (backend nccl = RCCL) and added in segment order."""
    lib = S.load_samplers()
    lib.gsl_ran_gaussian_ziggurat.restype = C.c_double
    lib.gsl_ran_gaussian_ziggurat.argtypes = [C.c_void_p, C.c_double]
    lib.gsl_rng_uniform.restype = C.c_double
    lib.gsl_rng_uniform.argtypes = [C.c_void_p]
    rng = S.Rng(a.seed)
    slots = torch.zeros(lk.segments, dtype=torch.float64, device=dev)

    def likelihood(theta):
        v = lk.loglik(theta)
        if world == 1:
            return v
        slots.copy_(torch.from_numpy(lk.segment_sums()))
        dist.all_reduce(slots, op=dist.ReduceOp.SUM)
        total = 0.0
        for x in slots.cpu().numpy():
            total += float(x)
        return total

    def log_prior(theta):
        return float(np.sum(np.log(np.exp(-(theta / 10.0) ** 2 / 2.0) / (np.sqrt(2.0 * np.pi) * 10.0))))

    cur = np.zeros(d)
    cpost = log_prior(cur) + likelihood(cur)
    acc = [0]
    ms = []

    def step(count):
        nonlocal cur, cpost
        prop = cur + np.array([lib.gsl_ran_gaussian_ziggurat(rng.r, rwsd) for _ in range(d)])
        L = likelihood(prop)
        ms.append(lk.last_kernel()[0])
        ppost = log_prior(prop) + L
        u = lib.gsl_rng_uniform(rng.r)
        if np.log(u) <= ppost - cpost:
            cur, cpost = prop, ppost
            if count:
                acc[0] += 1

    for _ in range(max(W - 1, 0)):
        step(False)
    ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, float(np.mean(ms)) if ms else 0.0, acc[0]


def workload_name(N, d):
    """Which BASELINE.json configuration these sizes are."""
    return {(1_000_000, 16): "BASELINE configs[1]", (10_000_000, 64): "BASELINE configs[2], the metric's N=10M d=64",
            (100_000_000, 64): "BASELINE configs[3]"}.get((N, d), "custom size")


TRAFFIC_PROFILE = "profiles/r03_bench_pmc_hbm_traffic.json"


def pmc_traffic(N, d, local_rows, kernel_name):
    """HBM bytes per launch of the streaming kernel from the committed rocprofv3
    PMC passes (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE;
    tools/profile_bench.sh).  Counters cannot be read from inside this process,
    so the figure is only reported when the committed passes were taken on this
    same workload AND on exactly the kernel instantiation this run just timed
    (mhlik_last_kernel_name); otherwise null.  Returns (bytes, source file)."""
    try:
        prof = json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE)))
    except (OSError, ValueError):
        return None, None
    w = prof.get("workload", {})
    if w.get("rows") != N or w.get("cols") != d or local_rows != N or not kernel_name:
        return None, None
    for name, k in prof.get("kernels", {}).items():
        # rocprofv3 spells it "void mhlik::lik_tile_kernel<double, 1, 16, true, false, false>(double const*, ...)"
        if name.split("(")[0].endswith("mhlik::" + kernel_name):
            return k.get("hbm_bytes_per_launch_corrected"), TRAFFIC_PROFILE
    return None, None


def extras(a, lk, S, np, N, d):
    """SP twin of the workload (BASELINE configs[2] is 'SP vs DP')."""
    from mcmc_amd import Likelihood, F32

    ex = {}
    K = min(a.steps, 200)
    with Likelihood(N, d, precision=F32, timing=True) as lf:
        lf.fill_synthetic(a.data_seed)
        with S.attached(lf):
            t0 = time.perf_counter()
            r = S.sp_sampler(None, a.seed, K, 2, 2.38 / np.sqrt(d) * 2e-3, trace=False, n_rows=N, n_cols=d)
            el = time.perf_counter() - t0
        kms = float(r["res"].kernelTime)
        ex["sp"] = {"it_per_s": (K + 2) / el, "kernel_ms": kms,
                    "GBps": N * d * 4 / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                    "frac_hbm_peak": N * d * 4 / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kms > 0 else 0.0}
        th = np.random.default_rng(3).normal(0, 0.01, d)
        dp = lk.loglik(th)
        sp = lf.loglik_f32(th.astype(np.float32))
        ex["sp"]["rel_diff_vs_dp"] = abs(sp - dp) / abs(dp)
    return ex


def cpu_baseline(a, np, N, d, rwsd, lk, S):
    """The oracle's cpu_sampler (restated mcmc_cpu, 1 thread like the
    reference) on a bounded sample of the SAME workload: the same N x d data,
    a short seeded chain.  The same short chain is run on the GPU and the two
    traces are compared (accept/reject sequence, log-likelihoods)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(a.data_seed, N, d)
    upload = host_upload_rate(X, N, d)
    iters, burn = a.cpu_iters, 2
    t0 = time.perf_counter()
    r = O.cpu_sampler(X, a.seed, iters, burn, rwsd)
    el = time.perf_counter() - t0
    evals = iters + burn  # likelihood evaluations = MH iterations incl. the chain start
    with S.attached(lk):
        g = S.gpu_sampler(None, a.seed, iters, burn, rwsd, n_rows=N, n_cols=d)
    ct, gt = r["trace"], g["trace"]
    rel = np.abs(ct["lik"] - gt["lik"]) / np.abs(ct["lik"])
    contested = contested_parity(a, np, O, S, lk, X, N, d, rwsd) if a.contested_iters > 0 else None
    L = O.lib()
    th = np.zeros(d)
    ncores = min(os.cpu_count() or 1, 16)
    t1 = time.perf_counter()
    L.oracle_log_likelihood_mt(O.ptr(th), O.ptr(X), N, d, N, ncores)
    mt = time.perf_counter() - t1
    return {"value": evals / el, "unit": "it/s", "cores": 1, "kind": "port",
            "sample": "%d MH iterations (%d sampling + %d burn-in) of the oracle cpu_sampler at the full "
                      "N=%d d=%d, same data and chain seed" % (evals, iters, burn, N, d),
            "seconds": el,
            "parity": {"iterations_compared": int(len(ct)),
                       "accept_trace_equal": bool(np.array_equal(ct["accept"], gt["accept"])),
                       "samples_equal": bool(np.array_equal(r["samples"], g["samples"])),
                       "max_rel_loglik_diff": float(rel.max()) if len(rel) else None,
                       "min_decision_margin": float(np.abs(ct["logu"] - ct["ratio"]).min()) if len(rel) else None,
                       "note": "this chain starts at theta = 0 with small steps: its decisions are not contested; "
                               "see `contested`",
                       "contested": contested},
            "all_cores": {"it_per_s_likelihood_only": 1.0 / mt, "cores": ncores},
            "host_upload": upload}


def contested_parity(a, np, O, S, lk, X, N, d, rwsd):
    """Accept/reject parity where it can fail: the chain is first walked on the
    GPU and its step size rescaled until a short run's median |log ratio| is a
    few nats (log u and the ratio are then of the same size, 15-60 % of the
    proposals accepted), then the SAME start and step size go to the
    oracle's cpu_sampler and to gpu_sampler (tests/compare_cpu_gpu.c:89-110
    shape, re-seeded) and the two traces are compared decision by decision."""
    with S.attached(lk):
        theta, sd, acc, _ = S.walk_to_contested(N, d, a.seed, np.zeros(d), rwsd)
        g = S.gpu_sampler(None, a.seed + 7, a.contested_iters, 1, sd, theta0=theta, n_rows=N, n_cols=d)
    r = O.cpu_sampler(X, a.seed + 7, a.contested_iters, 1, sd, theta0=theta)
    ct, gt = r["trace"], g["trace"]
    rel = np.abs(ct["lik"] - gt["lik"]) / np.abs(ct["lik"])
    margin = np.abs(ct["logu"] - ct["ratio"])
    worst = int(np.argmin(margin))
    return {"iterations_compared": int(len(ct)), "rwsd": float(sd), "walk_acceptance": acc,
            "acceptance": float(np.mean(ct["accept"])),
            "accept_trace_equal": bool(np.array_equal(ct["accept"], gt["accept"])),
            "samples_equal": bool(np.array_equal(r["samples"], g["samples"])),
            "median_abs_ratio": float(np.median(np.abs(ct["ratio"]))),
            "min_decision_margin": float(margin.min()),
            "abs_loglik_diff_at_min_margin": float(abs(ct["lik"][worst] - gt["lik"][worst])),
            "max_rel_loglik_diff": float(rel.max()), "max_abs_loglik_diff": float(np.abs(ct["lik"] - gt["lik"]).max())}


def host_upload_rate(X, N, d):
    """One-time cost outside the per-iteration path: mhlik_upload_f64 of the
    host column-major matrix (pageable memory -> tiled layout in HBM)."""
    from mcmc_amd import Likelihood, F64

    with Likelihood(N, d, precision=F64) as up:
        t0 = time.perf_counter()
        up.upload(X)
        el = time.perf_counter() - t0
    return {"seconds": el, "GBps": N * d * 8 / el / 1e9, "bytes": N * d * 8}


if __name__ == "__main__":
    main()
