#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Gsample/s (complex int16) of the batched N=1024
16/16 scaled DIF FFT (BASELINE config 2, batch 65536 per GPU), natural in -> natural out.

A "step" is one pass of the hot path (one intfft_exec) over one resident batch of synthetic frames.
Inputs live in HBM before the timed region; N > 1 shards the batch (independent frames, no data-path
collective) -> weak scaling, one process per GPU (torch.distributed / RCCL only for barrier + max).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
LOG2N = 10
N = 1 << LOG2N
BYTES_PER_SAMPLE = 8  # int16 (re, im) in + int16 (re, im) out, each touched once


def make_input(batch: int, rank: int):
    """Config 2 synthetic input: frames 0..7 = edge set, the rest i.i.d. uniform in [-2^14, 2^14)."""
    import numpy as np
    import torch

    from tests.helpers import edge_frames

    g = torch.Generator(device="cuda")
    g.manual_seed(0xC0FFEE02 + rank)
    x = torch.randint(-(1 << 14), 1 << 14, (batch, N, 2), device="cuda", dtype=torch.int16, generator=g)
    if batch >= 8:
        x[:8] = torch.from_numpy(edge_frames(N, 16).astype(np.int16)).cuda()
    return x


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/, collected by tools/profile.sh on this same command); None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_k_fft1024_pmc_digest.json")
    try:
        with open(path) as fh:
            return float(json.load(fh)["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(x_dev, y_dev):
    """Times the oracle (a scalar C port of the RTL arithmetic in the reference model's dataflow,
    OpenMP over frames) on a bounded sample of the same workload, and uses the result as a parity
    gate for the GPU output of those frames."""
    import numpy as np

    from oracle import oracle_c as C

    p = C.make_params(LOG2N, 16, 16, 0, 0, True)
    threads = C.num_threads()
    calib = 512
    xs = x_dev[:calib].cpu().numpy()
    C.execute_i16(xs, p, C.FWD, form=1, threads=threads)  # warm the OpenMP team
    t0 = time.perf_counter()
    C.execute_i16(xs, p, C.FWD, form=1, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-5)
    # aim at ~12 s of CPU work (thread-seconds), bounded by the batch
    frames = int(min(x_dev.shape[0], max(calib, calib * (12.0 / threads) / dt)))
    xs = np.ascontiguousarray(x_dev[:frames].cpu().numpy())
    ref = np.zeros_like(xs)  # pre-touched: page faults are not part of the baseline
    import ctypes
    t0 = time.perf_counter()
    rc = C.lib().orc_exec_i16(ctypes.byref(p), C.FWD, C.NATURAL, C.NATURAL, xs.ctypes.data, ref.ctypes.data,
                              frames, 1, threads)
    dt = time.perf_counter() - t0
    assert rc == 0
    got = y_dev[:frames].cpu().numpy()
    parity = bool(np.array_equal(got, ref))
    return {"value": frames * N / dt / 1e9, "unit": "Gsample/s", "cores": threads, "kind": "port",
            "sample": "%d frames of the same N=1024 16/16 scaled-truncate workload, oracle in-place form, "
                      "OpenMP over frames" % frames,
            "parity_checked_frames": frames, "parity_ok": parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU (config 2: 65536)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm", type=int, default=400,
                    help="untimed clock-ramp launches before the W warmup steps (the GPU needs ~300 "
                         "back-to-back launches to reach its steady shader clock; see DESIGN.md)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # INTFFT_BENCH_SHARE_GPU=1 (diagnostics only): several ranks on one GPU over gloo, to exercise the N > 1
    # control flow on a 1-GPU box; the driver's runs use one GPU per rank over RCCL.
    share = os.environ.get("INTFFT_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    from intfftk_amd import int_fft_single_path

    core = int_fft_single_path(NFFT=LOG2N, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=0, RNDMODE=0, device=local_rank)
    x = make_input(args.batch, rank)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    in_ptr, out_ptr = x.data_ptr(), y.data_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.prewarm):  # untimed: DVFS ramp, not part of W/K
        core.exec_raw(in_ptr, out_ptr, args.batch, stream)
    for _ in range(args.warmup):
        core.exec_raw(in_ptr, out_ptr, args.batch, stream)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        core.exec_raw(in_ptr, out_ptr, args.batch, stream)
    ev1.record()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    from intfftk_amd.sharding import max_over_ranks

    t_all = max_over_ranks(t_local, None if share else torch.device("cuda", local_rank))
    kern_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream

    if rank == 0:
        samples = float(args.batch) * N * world * args.steps
        launches = core.info["n_passes"]
        achieved = BYTES_PER_SAMPLE * args.batch * N / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "Gsample/s (complex int16) batched N=1024 scaled FFT",
            "value": samples / t_all / 1e9,
            "unit": "Gsample/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_all / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {"workload": "configs[1]: N=1024, 16-bit data / 16-bit twiddle, scaled-truncate DIF FFT, "
                                   "natural->natural, batch=%d per GPU" % args.batch,
                       "batch_per_gpu": args.batch, "n": N, "parallelism": "batch-shard x%d" % world,
                       "kernel": core.info["kernel_name"], "launches_per_step": launches,
                       "clock_prewarm_steps": args.prewarm},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic() if args.batch == 65536 else None,
                         "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * args.batch * N,
                         "kernel_ms": kern_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x, y)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
