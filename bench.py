#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Gsample/s (complex int16) of the batched N=1024
16/16 scaled DIF FFT (BASELINE config 2, batch 65536 per GPU), natural in -> natural out, with % of the
HBM roofline, at 1/2/4/8 GPUs.

A "step" is one pass of the hot path (one intfft_exec) over one resident batch of synthetic frames.
Inputs live in HBM before the timed region; N > 1 shards the batch (independent frames, NO data-path
collective) -> weak scaling, one process per GPU over RCCL (torch.distributed backend "nccl"), used for the
barrier, the max-over-ranks timing rule and -- with --e2e -- the root scatter / gather.

Launch forms:
  python bench.py                          1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N    (the driver's form)
  python bench.py --gpus N                 no WORLD_SIZE in the environment: spawns its own N ranks (127.0.0.1)
Options: --config C2|C3|C4|C5 (the other BASELINE configurations in the same line format: C3 = N=65536 24-bit unscaled, 4096 frames;
C4 = N=2^20 16-bit scaled, 1024 frames; C5 = N=4096 FFT->IFFT pair, 16384 frames per GPU), --e2e (adds the end-to-end
scatter -> transform -> gather rate as a separate field; never `value`).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`, plus the
section-8(d) side figures at N=1: cold-clock and full-scale-input rates, the on-box copy ceiling of the kernel's
access pattern, a VALU-issue bound, single-thread / stream-form CPU baselines, an Octave probe.
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
BYTES_PER_SAMPLE = 8   # int16 (re, im) in + int16 (re, im) out, each touched once

CONFIGS = {
    # name: (log2n, direction, frames per GPU, seed, workload text)
    "C2": (10, "FWD", 65536, 0xC0FFEE02,
           "configs[1]: N=1024, 16-bit data / 16-bit twiddle, scaled-truncate DIF FFT, natural->natural"),
    "C3": (16, "FWD", 4096, 0xC0FFEE03,
           "configs[2]: N=65536, 24-bit data / 24-bit twiddle, unscaled (full bit growth: 40-bit results) DIF FFT, natural->natural"),
    "C4": (20, "FWD", 1024, 0xC0FFEE04,
           "configs[3]: N=2^20, 16-bit data / 16-bit twiddle, scaled-truncate DIF FFT, Taylor twiddles (row_twiddle_tay), natural->natural"),
    "C5": (12, "PAIR", 16384, 0xC0FFEE05,
           "configs[4]: N=4096, 16-bit scaled FFT->IFFT pair (int_fft_ifft_pair), natural->natural, batch split over the GPUs"),
}
# generics and accounting per config: (DATA_WIDTH, TWDL_WIDTH, FORMAT, algorithmic bytes per sample, dtype of the line, metric)
GENERICS = {
    "C2": (16, 16, 0, 8, "int16", "Gsample/s (complex int16) batched N=1024 scaled FFT"),
    "C3": (24, 24, 1, 24, "int64", "Gsample/s (complex, 24-bit in int32 -> 40-bit in int64) batched N=65536 unscaled FFT"),
    "C4": (16, 16, 0, 8, "int16", "Gsample/s (complex int16) batched N=2^20 scaled FFT (Taylor twiddles)"),
    "C5": (16, 16, 0, 8, "int16", "Gsample/s (complex int16) batched N=4096 scaled FFT->IFFT pair"),
}


def make_input(batch: int, n: int, seed: int, rank: int, full_scale: bool = False, dw: int = 16):
    """Synthetic input: frames 0..7 = edge set, the rest i.i.d. uniform in [-2^(w-2), 2^(w-2))
    (full_scale: the whole w-bit range -- every frame then fails the guard-bit vote and takes the exact extraction)."""
    import numpy as np
    import torch

    from tests.helpers import edge_frames

    g = torch.Generator(device="cuda")
    g.manual_seed(seed + rank)
    lim = 1 << (dw - 1 if full_scale else dw - 2)
    dt, npdt = (torch.int16, np.int16) if dw <= 16 else (torch.int32, np.int32)
    x = torch.randint(-lim, lim, (batch, n, 2), device="cuda", dtype=dt, generator=g)
    if batch >= 8 and not full_scale:
        x[:8] = torch.from_numpy(edge_frames(n, dw).astype(npdt)).cuda()
    return x


def pmc_digest(config="C2"):
    """The newest committed rocprofv3 PMC digest of this configuration's kernel(s): (path, {"hbm_bytes_per_launch", "SQ_INSTS_VALU",
    "SQ_WAVES", ...}) or (None, None).  profiles/rNN_<config>_pmc_digest.json (tools/pmc_digest.sh: one --pmc pass per counter set
    over tools/bench_configs.py <config>, the same plan / batch as this bench) -- per-call HBM bytes = sum over the plan's kernels;
    older rounds: profiles/rNN_k_fft1024_pmc_digest.json (tools/profile.sh on bench.py itself, C2 only)."""
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_digest.json" % config)))):
        try:
            with open(path) as fh:
                d = json.load(fh)
            k = max(d["kernels"].values(), key=lambda v: v.get("avg_ns", 0.0))  # the dominant kernel
            out = dict(k)
            out["hbm_bytes_per_launch"] = d["hbm_bytes_per_call"]
            out["traffic_over_algorithmic"] = d.get("traffic_over_algorithmic")
            # VALU wave-instructions of ONE intfft_exec call: every kernel of the plan x its launches per call
            out["valu_insts_per_call"] = sum(float(v.get("SQ_INSTS_VALU", 0.0)) * float(v.get("launches_per_call", 1.0))
                                             for v in d["kernels"].values())
            out["digest_batch"] = d.get("batch")
            out["lib_version"] = d.get("lib_version")
            return os.path.relpath(path, ROOT), out
        except Exception:
            continue
    if config != "C2":
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k_fft1024_pmc_digest.json")))
    for path in reversed(files):
        try:
            with open(path) as fh:
                return os.path.relpath(path, ROOT), json.load(fh)
        except Exception:
            continue
    return None, None


def loaded_lib_version():
    from intfftk_amd import _capi

    return _capi.lib().intfft_version().decode()


def digest_stale(digest):
    """True when the committed PMC digest was NOT taken on the sources of the library this run measures (intfft_version() carries their
    hash): `traffic` and the PMC-based VALU bound are then figures of an older build and say so."""
    return bool(digest) and digest.get("lib_version") != loaded_lib_version()


# ---- CPU baselines (the oracle = a scalar C port of the RTL arithmetic in the reference model's dataflow) ----------
def cpu_baseline(x_dev, y_dev, log2n, direction, dw=16, tw=16, fmt=0, quick=False, rnd=0):
    """(quick: one form -- the stream form -- on all host threads over a smaller fixed sample, one timed pass after the warm one; the
    compact baseline / parity gate of the `other_configs` sub-records.)
    Times the oracle on FIXED samples of the same workload (no adaptive sizing: the figures are comparable from run
    to run) and uses the all-core passes as parity gates for the GPU output of those frames.  Headline entry (`value`): the
    FASTER of the two forms on all host threads -- the stream form (the literal dataflow of math/fn_radix2.m: half-split
    lanes, per-stage butterflies, fn_rev2rdx commutation) or the flat in-place form; `form` says which.  Both forms are
    also reported on their own, all-core and single-thread."""
    import numpy as np

    from oracle import oracle_c as C

    n = 1 << log2n
    p = C.make_params(log2n, dw, tw, fmt, rnd, True)
    d = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}[direction]
    i16 = dw <= 16 and fmt == 0  # int16 containers both ways: the oracle's int16 entry point; else its int64 one
    threads = C.num_threads()
    scale = 1024 // n if n <= 1024 else 1
    work = (2 if direction == "PAIR" else 1) * max(1, n // 1024)

    def timed(frames, form, nthreads, reps=3):
        frames = max(8, min(max(int(frames), nthreads), x_dev.shape[0]))  # (at least one frame per thread: long frames)
        xs = np.ascontiguousarray(x_dev[:frames].cpu().numpy() if i16 else x_dev[:frames].cpu().numpy().astype(np.int64))
        ref = np.zeros_like(xs)  # pre-touched: page faults are not part of the baseline
        fn = C.lib().orc_exec_i16 if i16 else C.lib().orc_exec
        best = None
        for _ in range(reps + 1):  # first pass warms the OpenMP team and the caches
            t0 = time.perf_counter()
            rc = fn(ctypes.byref(p), d, C.NATURAL, C.NATURAL, xs.ctypes.data, ref.ctypes.data, frames, form, nthreads)
            dt = time.perf_counter() - t0
            assert rc == 0
            best = dt if best is None else min(best, dt)
        return frames, frames * n / best / 1e9, ref

    if quick:
        f_q, v_q, ref_q = timed(8192 * scale // work, 0, threads, reps=1)
        return {"value": v_q, "unit": "Gsample/s", "cores": threads, "kind": "port", "form": "stream",
                "sample": "first %d frames of the same workload, stream form, OpenMP over frames, one timed pass after a warm one" % f_q,
                "parity_checked_frames": f_q, "parity_ok": bool(np.array_equal(y_dev[:f_q].cpu().numpy(), ref_q))}
    # fixed sample sizes: ~10-20 s of CPU thread-time in total on a 128-thread host
    f_sall, v_sall, ref_s = timed(32768 * scale // work, 0, threads)
    parity_stream = bool(np.array_equal(y_dev[:f_sall].cpu().numpy(), ref_s))
    f_sone, v_sone, _ = timed(1024 * scale // work, 0, 1, reps=2)
    f_all, v_all, ref = timed(32768 * scale // work, 1, threads)
    parity = bool(np.array_equal(y_dev[:f_all].cpu().numpy(), ref))
    f_one, v_one, _ = timed(1024 * scale // work, 1, 1, reps=2)
    best_form = "stream" if v_sall >= v_all else "in_place"
    return {"value": max(v_sall, v_all), "unit": "Gsample/s", "cores": threads, "kind": "port", "form": best_form,
            "sample": "first %d frames of the same workload, the faster of the oracle's two forms (here: %s), OpenMP over "
                      "frames, best of 3 (fixed sample)" % (f_sall, best_form),
            "parity_checked_frames": f_sall, "parity_ok": parity_stream and parity,
            "stream_form": {"value": v_sall, "cores": threads, "sample": "%d frames, stream form (the dataflow of math/fn_radix2.m "
                                                                         "with the RTL's integer butterflies)" % f_sall,
                            "parity_ok": parity_stream},
            "single_thread": {"value": v_sone, "cores": 1, "sample": "%d frames, stream form" % f_sone},
            "in_place_form": {"value": v_all, "cores": threads, "sample": "%d frames, flat in-place form" % f_all,
                              "parity_ok": parity},
            "in_place_form_single_thread": {"value": v_one, "cores": 1, "sample": "%d frames, flat in-place form" % f_one}}


def octave_probe():
    """the reference model as shipped, if Octave exists on this box.  The model's .m files are
    reference sources and do not travel with this repository; point INTFFT_REFERENCE_DIR at a checkout to time it."""
    exe = shutil.which("octave") or shutil.which("octave-cli")
    if not exe:
        return {"available": False, "note": "octave: not available on this host"}
    ref = os.environ.get("INTFFT_REFERENCE_DIR", "")
    if not ref or not os.path.isdir(os.path.join(ref, "math")):
        return {"available": True, "note": "octave found; set INTFFT_REFERENCE_DIR to a checkout of hukenovs/intfftk to time "
                                           "math/fn_radix2.m (reference sources are not shipped here)"}
    prog = ("addpath('%s'); pkg load signal; N=1024; i=(0:N-1)'; ph=(24*i+0.95*i.*i/2)*2*pi/N; w=sin(i*pi/N);"
            "Din=round(255*cos(ph).*w)+1j*round(255*sin(ph).*w); fn_radix2(Din,N,'FWD'); tic; for k=1:10,"
            "fn_radix2(Din,N,'FWD'); end; printf('%%g', toc/10);" % os.path.join(ref, "math"))
    try:
        out = subprocess.run([exe, "--no-gui", "--quiet", "--eval", prog], capture_output=True, text=True, timeout=120)
        sec = float(out.stdout.strip().split()[-1])
        return {"available": True, "seconds_per_frame": sec, "value": 1024 / sec / 1e9, "unit": "Gsample/s",
                "sample": "fn_radix2(Din, 1024, 'FWD') on the C1 chirp, mean of 10"}
    except Exception as exc:  # diagnostics only
        return {"available": True, "note": "octave run failed: %r" % (exc,)}


# ---- on-box ceilings (tools/lib/libintfft_diag.so; diagnostics, never on the measured path) ------------------------
def diag_lib():
    path = os.path.join(ROOT, "tools", "lib", "libintfft_diag.so")
    if not os.path.exists(path):
        return None
    L = ctypes.CDLL(path)
    L.diag_copy_wave_nt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    if hasattr(L, "diag_copy_wave_ld"):
        L.diag_copy_wave_ld.argtypes = L.diag_copy_wave_nt.argtypes
    L.diag_valu_chain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_void_p]
    if hasattr(L, "diag_body"):
        L.diag_body.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_void_p]
    return L


def event_ms(torch, fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ceilings(torch, x, y, stream, digest, c2, samples_per_call=None, copy=True):
    L = diag_lib()
    if L is None:
        return {"note": "tools/lib/libintfft_diag.so not built"}
    out = {}
    if not copy:
        return valu_rates(torch, L, stream, digest, c2, samples_per_call, out)
    nframes = x.numel() * x.element_size() // 4096
    fn = L.diag_copy_wave_ld if hasattr(L, "diag_copy_wave_ld") else L.diag_copy_wave_nt
    copy = lambda: fn(x.data_ptr(), y.data_ptr(), nframes, 4, stream)  # noqa: E731
    for _ in range(50):
        copy()
    ms = event_ms(torch, copy, 50)
    gbs = 2.0 * nframes * 4096 / (ms * 1e-3) / 1e9
    out["copy_ceiling"] = {"GB/s": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "ms": ms,
                           "what": "copy of the same buffers with the kernel's access pattern (one wave per 4 KiB frame, 16 plain dword "
                                   "loads + 16 non-temporal dword stores per lane), measured in this run"}
    if fn is not L.diag_copy_wave_nt:  # the pattern of rounds 1-2, for comparison
        copy2 = lambda: L.diag_copy_wave_nt(x.data_ptr(), y.data_ptr(), nframes, 4, stream)  # noqa: E731
        for _ in range(50):
            copy2()
        ms2 = event_ms(torch, copy2, 50)
        out["copy_ceiling"]["nt_loads_too_GB/s"] = 2.0 * nframes * 4096 / (ms2 * 1e-3) / 1e9
    return valu_rates(torch, L, stream, digest, c2, samples_per_call, out)


def valu_rates(torch, L, stream, digest, c2, samples_per_call, out):
    """The VALU-issue bound of this configuration: the slow-class issue rate measured in this run (v_pk_*, v_dot2_i32_i16, v_perm_b32,
    v_bfe_i32, v_mad_i64_i32 ... all issue at ~4.4 clk per wave-instruction on gfx950) over the plan's SQ_INSTS_VALU per call from
    the committed PMC digest of the same command."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    scratch = torch.empty(cus * 4 * 256, dtype=torch.int32, device="cuda")
    n = ctypes.c_ulonglong()
    rates = {}
    for slow in (1, 0):
        chain = lambda: L.diag_valu_chain(slow, 2000, scratch.data_ptr(), ctypes.byref(n), stream)  # noqa: E731
        for _ in range(3):
            chain()
        ms = event_ms(torch, chain, 5)
        rates[slow] = n.value / (ms * 1e-3)  # wave-instructions per second, whole chip, 4 waves per SIMD
    clk = torch.cuda.get_device_properties(0).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2.4e9
    insts = None
    if c2 and digest and digest.get("SQ_INSTS_VALU") and digest.get("SQ_WAVES"):
        insts = float(digest["SQ_INSTS_VALU"]) / 65536.0  # VALU wave-instructions per 1024-sample frame (PMC, per launch / frames)
    vb = {"slow_class_wave_insts_per_s": rates[1], "fast_class_wave_insts_per_s": rates[0],
          "slow_class_clk_per_wave_inst_at_nominal_clock": cus * 4 * clk / rates[1],
          "what": "issue rate of v_pk_add_u16 (the class of v_pk_*, v_dot2_i32_i16, v_perm_b32, v_bfe_i32) and of v_add_u32, "
                  "4 waves per SIMD on every CU, measured in this run"}
    if insts:
        vb["valu_insts_per_frame_wave"] = insts
        vb["value"] = rates[1] / insts * 1024 / 1e9
        vb["unit"] = "Gsample/s"
        vb["note"] = "bound if every VALU instruction of the kernel (SQ_INSTS_VALU of the committed PMC digest) issued at the slow-class rate"
    elif (not c2) and digest and digest.get("valu_insts_per_call") and samples_per_call:
        vb["valu_insts_per_call"] = digest["valu_insts_per_call"]
        vb["value"] = rates[1] / float(digest["valu_insts_per_call"]) * samples_per_call / 1e9
        vb["unit"] = "Gsample/s"
        vb["note"] = ("bound if every VALU instruction of the plan's kernels (sum of SQ_INSTS_VALU x launches per call, committed PMC "
                      "digest) issued at the slow-class rate")
    out["valu_bound"] = vb
    return out


# ---- BASELINE's other configurations on the driver's clock (default single-GPU run) --------------------------------
def other_config(torch, name, steps, warmup, dev_index, slow_rate, cpu=True):
    """One sub-record of `other_configs`: `steps` calls of the C3 / C4 / C5 plan in THIS process, timed like the headline (wall clock
    between synchronisations = `ms_per_step`, HIP events on the launch stream = `kernel_ms`), the roofline fields of `--config <name>`
    (bounds from the committed PMC digest of the same plan and the slow-class VALU issue rate measured in this run), and the oracle
    on a bounded prefix of the same input as CPU baseline and parity gate.  Buffers are freed before returning (C4: 8 GiB)."""
    from intfftk_amd import int_fft_ifft_pair, int_fft_single_path

    log2n, direction, batch, seed, workload = CONFIGS[name]
    dw, tw, fmt, bps, line_dtype, metric = GENERICS[name]
    n = 1 << log2n
    ctor = int_fft_single_path if direction == "FWD" else int_fft_ifft_pair
    core = ctor(NFFT=log2n, DATA_WIDTH=dw, TWDL_WIDTH=tw, FORMAT=fmt, RNDMODE=0, device=dev_index)
    x = make_input(batch, n, seed, 0, dw=dw)
    y = torch.empty(core.out_shape(batch), device=x.device, dtype=core.out_dtype)
    stream = torch.cuda.current_stream().cuda_stream
    in_ptr, out_ptr = x.data_ptr(), y.data_ptr()
    step = lambda: core.exec_raw(in_ptr, out_ptr, batch, stream)  # noqa: E731
    step()
    torch.cuda.synchronize()
    ramp = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:  # untimed clock ramp (the CPU legs before this leave the GPU idle), then W warm-up steps
        for _ in range(5):
            step()
        ramp += 5
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps
    alg_bytes = float(bps) * batch * n
    here = batch * n / kern_ms / 1e6
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    digest_path, digest = pmc_digest(name)
    same = bool(digest) and digest.get("digest_batch") in (None, batch)
    traffic = float(digest["hbm_bytes_per_launch"]) if same else None
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "frac_hbm": achieved / HBM_PEAK_GBS, "traffic": traffic,
         "traffic_source": ("%s (static)" % digest_path) if traffic else None, "algorithmic_bytes_per_step": alg_bytes}
    if traffic:
        r["traffic_stale"] = digest_stale(digest)
    bounds = {"hbm": HBM_PEAK_GBS / bps}
    if traffic:
        bounds["hbm_pass_traffic"] = HBM_PEAK_GBS / (traffic / (float(batch) * n))
        r["frac_hbm_pass_traffic"] = here / bounds["hbm_pass_traffic"]
        r["traffic_over_algorithmic"] = traffic / alg_bytes
    if same and slow_rate and digest.get("valu_insts_per_call"):
        bounds["valu"] = slow_rate / float(digest["valu_insts_per_call"]) * float(batch) * n / 1e9
        r["frac_valu"] = here / bounds["valu"]
        r["bound"] = "valu" if bounds["valu"] < min(v for k, v in bounds.items() if k != "valu") else "hbm"
    r["bounds_Gsample_per_s"] = bounds
    out = {"workload": "%s, batch=%d" % (workload, batch), "metric": metric, "dtype": line_dtype, "value": batch * n * steps / wall / 1e9,
           "unit": "Gsample/s", "steps": steps, "warmup": warmup, "clock_ramp_steps": ramp, "ms_per_step": wall / steps * 1e3,
           "kernel_ms": kern_ms, "kernel": core.info["kernel_name"], "launches_per_step": core.info["n_passes"], "roofline": r}
    if cpu:
        out["cpu_baseline"] = cpu_baseline(x, y, log2n, direction, dw, tw, fmt, quick=True)
    core.close()
    del x, y, core
    torch.cuda.empty_cache()
    return out


# ---- the rest of the north-star shape on the driver's clock: N = 1024, 16/16 scaled, batch 65536 in every mode -----------------
# (the reference testbench instantiates the three modes side by side, src/vhdl/tb/fft_signle_test.vhd:80-112; DIF rounding
# int_dif2_fly.vhd:167-219, DIT rounding int_dit2_fly.vhd:164-217)
MODES = {
    # name: (RNDMODE, direction, text)
    "round_fwd": (1, "FWD", "RNDMODE=1 (round half up) DIF FFT"),
    "round_inv": (1, "INV", "RNDMODE=1 DIT IFFT"),
    "round_pair": (1, "PAIR", "RNDMODE=1 FFT->IFFT pair (int_fft_ifft_pair)"),
    "trunc_inv": (0, "INV", "RNDMODE=0 (truncate) DIT IFFT"),
    "trunc_pair": (0, "PAIR", "RNDMODE=0 FFT->IFFT pair (int_fft_ifft_pair)"),
}
# VALU-issue floor of one N = 1024 frame per wave, in "wave-rounds" of the register-only butterfly bodies that tools/diag_kernels.hip
# (diag_body) times in this run: (kind, round, fastx) -> rounds per frame.  kind 0 / 2 = four general DIF / DIT stages on 16 registers
# (the frame has six: 1.5 rounds), kind 1 / 3 = stages 3, 2 with wave-uniform twiddles + STAGE 1 + STAGE 0 (one round).  fastx 1 = fast
# extraction (frames that pass the magnitude vote), 2 = the t = 16 exact extraction (phase 1 of a frame that fails it), 0 = v_bfe.
# The floor leaves out loads, stores, the LDS transpose, the lane swaps and the votes: it is an UPPER bound on the rate VALU issue allows.
BODY_MIX = {
    ("FWD", 0, False): {(0, 0, 1): 1.5, (1, 0, 1): 1.0},
    ("FWD", 0, True): {(0, 0, 2): 1.0, (0, 0, 1): 0.5, (1, 0, 1): 1.0},   # full-scale input: stages 9..6 exact, then the second vote passes
    ("FWD", 1, False): {(0, 1, 0): 1.5, (1, 1, 0): 1.0},
    ("INV", 0, False): {(2, 0, 1): 1.5, (3, 0, 1): 1.0},
    ("INV", 0, True): {(2, 0, 0): 1.5, (3, 0, 0): 1.0},                   # (the inverse wave kernel has one vote: a failing frame is exact throughout)
    ("INV", 1, False): {(2, 1, 0): 1.5, (3, 1, 0): 1.0},
}
BODY_MIX[("FWD", 1, True)] = BODY_MIX[("FWD", 1, False)]  # round mode has no fast extraction: the input distribution changes nothing
BODY_MIX[("INV", 1, True)] = BODY_MIX[("INV", 1, False)]
for _r in (0, 1):
    for _fs in (False, True):
        _m = dict(BODY_MIX[("FWD", _r, _fs)])
        for _k, _v in BODY_MIX[("INV", _r, False if _r == 0 and _fs else _fs)].items():  # (the forward half of a pair scales the data: the inverse half passes its vote)
            _m[_k] = _m.get(_k, 0.0) + _v
        BODY_MIX[("PAIR", _r, _fs)] = _m


def body_rates(torch, stream):
    """wave-rounds per second (whole chip, 4 waves per SIMD) of every butterfly body of BODY_MIX, measured now; {} without the diag library"""
    L = diag_lib()
    if L is None or not hasattr(L, "diag_body"):
        return {}
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    seed = torch.randint(-(1 << 31), (1 << 31) - 1, (256 * 48,), dtype=torch.int32, device="cuda")
    scratch = torch.empty(cus * 4 * 256, dtype=torch.int32, device="cuda")
    n = ctypes.c_ulonglong()
    rates = {}
    for key in sorted({k for mix in BODY_MIX.values() for k in mix}):
        fn = lambda: L.diag_body(key[0], key[1], key[2], 400, seed.data_ptr(), scratch.data_ptr(), ctypes.byref(n), stream)  # noqa: E731,B023
        if fn() != 0:
            continue
        for _ in range(3):
            fn()
        ms = event_ms(torch, fn, 5)
        rates[key] = n.value / (ms * 1e-3)
    return rates


def valu_floor(rates, direction, rnd, full_scale):
    """Gsample/s that VALU issue alone allows for N = 1024 frames of this mode (None if a body was not measured)"""
    mix = BODY_MIX[(direction, rnd, full_scale)]
    if any(k not in rates for k in mix):
        return None
    return 1024.0 / sum(c / rates[k] for k, c in mix.items()) / 1e9


def mode_roofline(here, kern_ms, alg_bytes, floor):
    r = {"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    r["frac_hbm"] = r["frac"]
    bounds = {"hbm": HBM_PEAK_GBS / BYTES_PER_SAMPLE}
    if floor:
        bounds["valu_bodies"] = floor
        r["frac_valu"] = here / floor
        r["bound"] = "valu" if floor < bounds["hbm"] else "hbm"
    r["bounds_Gsample_per_s"] = bounds
    return r


def other_mode(torch, name, steps, warmup, dev_index, rates, cpu=True):
    """One sub-record of `other_modes`: the C2 shape (N = 1024, 16-bit data / 16-bit twiddles, scaled, 65536 frames, natural -> natural)
    in another rounding mode / direction, timed like the headline; `roofline.frac_valu` = measured rate over the VALU-issue floor built
    from the butterfly bodies timed in this run (BODY_MIX); `full_scale_input` = the same with input uniform over the whole int16 range;
    the oracle on a bounded prefix is the parity gate."""
    from intfftk_amd import IntFFTCore

    rnd, direction, text = MODES[name]
    log2n, batch, n = 10, 65536, 1024
    core = IntFFTCore(log2n, 16, 16, 0, rnd, "NEW", direction, "NATURAL", "NATURAL", device=dev_index)
    x = make_input(batch, n, 0xC0FFEE02, 0)
    y = torch.empty(core.out_shape(batch), device=x.device, dtype=core.out_dtype)
    stream = torch.cuda.current_stream().cuda_stream
    in_ptr, out_ptr = x.data_ptr(), y.data_ptr()
    step = lambda: core.exec_raw(in_ptr, out_ptr, batch, stream)  # noqa: E731
    step()
    torch.cuda.synchronize()
    ramp = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:  # untimed clock ramp, then W warm-up steps
        for _ in range(20):
            step()
        ramp += 20
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps
    alg_bytes = float(BYTES_PER_SAMPLE) * batch * n
    here = batch * n / kern_ms / 1e6
    out = {"workload": "N=1024, 16-bit data / 16-bit twiddle, scaled, %s, natural->natural, batch=%d" % (text, batch),
           "dtype": "int16", "value": batch * n * steps / wall / 1e9, "unit": "Gsample/s", "steps": steps, "warmup": warmup,
           "clock_ramp_steps": ramp, "ms_per_step": wall / steps * 1e3, "kernel_ms": kern_ms, "kernel": core.info["kernel_name"],
           "launches_per_step": core.info["n_passes"],
           "roofline": mode_roofline(here, kern_ms, alg_bytes, valu_floor(rates, direction, rnd, False))}
    if cpu:
        c = cpu_baseline(x, y, log2n, direction, 16, 16, 0, quick=True, rnd=rnd)
        out["cpu_baseline"] = c
        out["parity_ok"] = c["parity_ok"]
    xf = make_input(batch, n, 0xC0FFEE02, 0, full_scale=True)
    stepf = lambda: core.exec_raw(xf.data_ptr(), out_ptr, batch, stream)  # noqa: E731
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:  # the oracle leg above left the GPU idle: ramp the clocks again before timing
        for _ in range(20):
            stepf()
        torch.cuda.synchronize()
    ms = event_ms(torch, stepf, steps)
    fs = {"kernel_ms": ms, "value": batch * n / ms / 1e6,
          "roofline": mode_roofline(batch * n / ms / 1e6, ms, alg_bytes, valu_floor(rates, direction, rnd, True))}
    if cpu:  # parity of the exact-extraction paths too: the first 2048 full-scale frames against the oracle
        import numpy as np

        from oracle import oracle_c as C

        torch.cuda.synchronize()
        want = C.execute(xf[:2048].cpu().numpy().astype(np.int64), C.make_params(log2n, 16, 16, 0, rnd, True),
                         {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}[direction])
        fs["parity_checked_frames"] = 2048
        fs["parity_ok"] = bool(np.array_equal(y[:2048].cpu().numpy().astype(np.int64), want))
    out["full_scale_input"] = fs
    core.close()
    del x, y, xf, core
    torch.cuda.empty_cache()
    return out


# ---- self-spawn: `python bench.py --gpus N` without a launcher -----------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), INTFFT_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU (default: the config's)")
    ap.add_argument("--e2e", action="store_true", help="also time root scatter -> transform -> gather (reported apart)")
    ap.add_argument("--rank-seeded-data", action="store_true",
                    help="every rank draws its own input (seed + rank); default: every rank transforms the SAME synthetic batch as "
                         "rank 0, so that the N = 1 line of a scaling run is the single-GPU bench line exactly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the section-8(d) side figures (profiling runs)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the C3 / C4 / C5 sub-records the default single-GPU C2 run appends as `other_configs`")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the `other_modes` sub-records (N = 1024 16-bit scaled in round mode / inverse / pair, each with its bounds)")
    ap.add_argument("--prewarm", type=int, default=400,
                    help="untimed clock-ramp launches before the W warmup steps (the GPU needs ~300 "
                         "back-to-back launches to reach its steady shader clock; see DESIGN.md)")
    args = ap.parse_args()

    # dmabuf IPC (the host driver has no legacy IPC): needed by RCCL between ranks; set here too, not only in spawn_ranks, so that
    # the driver's `python -m torch.distributed.run ... bench.py` path has it even if its environment dropped the variable
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))  # one process per GPU, rendezvous on 127.0.0.1

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    ndev = torch.cuda.device_count()
    # INTFFT_BENCH_SHARE_GPU=1 (diagnostics only): several ranks on one GPU over gloo, to exercise the N > 1
    # control flow on a 1-GPU box (RCCL refuses two ranks on one device); the driver's runs use one GPU per rank.
    share = os.environ.get("INTFFT_BENCH_SHARE_GPU") == "1" and ndev < world
    if ndev < world and not share:
        raise SystemExit("bench.py: %d ranks but only %d HIP device(s) visible (INTFFT_BENCH_SHARE_GPU=1 runs the control "
                         "flow on fewer devices over gloo, for diagnostics)" % (world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if share else "nccl"
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    elif args.e2e:  # a one-rank RCCL group so that the scatter / gather code path is the same at N = 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        backend = "nccl"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    red_dev = None if (share or world == 1) else device  # where the tiny control tensors live

    from intfftk_amd import int_fft_ifft_pair, int_fft_single_path
    from intfftk_amd.sharding import ShardedTransform, gather_floats, max_over_ranks

    log2n, direction, cfg_batch, seed, workload = CONFIGS[args.config]
    n = 1 << log2n
    batch = args.batch or cfg_batch
    dw, tw, fmt, bytes_per_sample, line_dtype, metric = GENERICS[args.config]
    ctor = int_fft_single_path if direction == "FWD" else int_fft_ifft_pair
    core = ctor(NFFT=log2n, DATA_WIDTH=dw, TWDL_WIDTH=tw, FORMAT=fmt, RNDMODE=0, device=dev_index)
    x = make_input(batch, n, seed, rank if args.rank_seeded_data else 0, dw=dw)
    y = torch.empty(core.out_shape(batch), device=x.device, dtype=core.out_dtype)
    stream = torch.cuda.current_stream().cuda_stream
    in_ptr, out_ptr = x.data_ptr(), y.data_ptr()
    step = lambda: core.exec_raw(in_ptr, out_ptr, batch, stream)  # noqa: E731
    alg_bytes = bytes_per_sample * batch * n

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rccl_ranks = None
    if world > 1:  # proves the collective layer really spans `world` device ranks
        t = torch.ones(1, device=device if not share else "cpu")
        dist.all_reduce(t)
        rccl_ranks = int(t.item())

    # cold figure: the first K launches from idle clocks (one untimed launch loads the code object)
    extras = world == 1 and not args.no_extras
    cold = None
    if extras:
        step()
        torch.cuda.synchronize()
        time.sleep(0.5)
        ms = event_ms(torch, step, args.steps)
        cold = {"kernel_ms": ms, "value": batch * n / ms / 1e6, "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "what": "first %d launches from idle clocks, no pre-warm" % args.steps}

    import gc

    gc.collect()  # BEFORE the clock ramp: a collection takes ~20 ms, and a GPU left idle that long drops its clocks again
    gc.disable()  # no collector pause inside the K timed steps (a 50-step region is only ~5 ms long)
    for _ in range(args.prewarm):  # untimed: DVFS ramp, not part of W/K
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        dist.barrier()
    t_all = max_over_ranks(t_local, red_dev)
    kern_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    per_gpu_ms = gather_floats(kern_ms, red_dev)

    # end-to-end: the whole batch on rank 0 -> one grouped scatter -> transform -> one grouped gather -> rank 0
    e2e = None
    if args.e2e and args.config in ("C2", "C5"):  # (int16 both ways: the per-peer byte accounting below)
        total = batch * world
        sh = ShardedTransform(core, n, core.in_dtype, core.out_dtype, device, stage_via_cpu=share)
        root_x = make_input(total, n, seed, 0) if rank == 0 else None
        times = []
        for it in range(3 + 5):
            barrier()
            t0 = time.perf_counter()
            res = sh.run_from_root(root_x, total, 0)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            if it >= 3:
                times.append(time.perf_counter() - t0)
        t_e2e = max_over_ranks(sum(times) / len(times), red_dev)
        groups_serial = sh.last_group_sizes[-2:]
        # the same end to end as a pipeline: shards in 4 pieces, the gather of piece k in ONE group with the scatter of piece k + 1
        # (full-duplex links), the root's own shard beside the first group
        times_p = []
        for it in range(2 + 5):
            barrier()
            t0 = time.perf_counter()
            res_p = sh.run_from_root_pipelined(root_x, total, 0, pieces=4)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            if it >= 2:
                times_p.append(time.perf_counter() - t0)
        t_pipe = max_over_ranks(sum(times_p) / len(times_p), red_dev)
        ok_p = bool(torch.equal(res_p, res)) if rank == 0 else None
        del res_p
        ok = None
        if rank == 0:  # same rows as the resident transform of rank 0's own shard
            ok = bool(torch.equal(res[:batch], core(root_x[:batch])))
        # scatter and gather timed on their own (max over ranks): the root drives world - 1 peers at once, one xGMI link each
        link = None
        if world > 1:
            res_local = sh.scatter(root_x, total, 0)
            ts, tg = [], []
            for it in range(2 + 3):
                barrier()
                t0 = time.perf_counter()
                loc = sh.scatter(root_x, total, 0)
                torch.cuda.synchronize()
                dist.barrier()
                t1 = time.perf_counter()
                sh.gather(res_local, total, 0)
                torch.cuda.synchronize()
                dist.barrier()
                t2 = time.perf_counter()
                if it >= 2:
                    ts.append(t1 - t0)
                    tg.append(t2 - t1)
            t_s, t_g = max_over_ranks(min(ts), red_dev), max_over_ranks(min(tg), red_dev)
            peer_bytes = (BYTES_PER_SAMPLE // 2) * batch * n
            link = {"scatter_ms": t_s * 1e3, "gather_ms": t_g * 1e3, "scatter_link_GBps": peer_bytes / t_s / 1e9,
                    "gather_link_GBps": peer_bytes / t_g / 1e9, "peers": world - 1,
                    "what": "bytes to / from ONE peer over the time of the whole grouped scatter / gather (all peers concurrently)"}
            del loc, res_local
        e2e = {"value": total * n / t_e2e / 1e9, "unit": "Gsample/s", "ms": t_e2e * 1e3, "frames": total,
               "bytes_moved_per_peer": (BYTES_PER_SAMPLE // 2) * batch * n, "link": link,
               "link_GBps": None if link is None else min(link["scatter_link_GBps"], link["gather_link_GBps"]),
               "mode": "root scatter -> transform -> gather, one batch_isend_irecv group each way "
                       "(%s)" % ("gloo via host staging: diagnostics" if share else "RCCL: ncclGroupStart/ncclSend,Recv/ncclGroupEnd"),
               "p2p_ops_per_group": groups_serial, "matches_resident": ok,
               "pipelined": {"value": total * n / t_pipe / 1e9, "unit": "Gsample/s", "ms": t_pipe * 1e3, "pieces": 4,
                             "speedup_over_serial": t_e2e / t_pipe, "matches_serial": ok_p,
                             "mode": "5 groups: step t = scatter of piece t + gather of piece t - 1 in one ncclGroupStart .. ncclGroupEnd; "
                                     "cannot show a gain on one GPU (no peers: the groups are empty)"}}
        del root_x, res

    if rank == 0:
        samples = float(batch) * n * world * args.steps
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        digest_path, digest = pmc_digest(args.config)
        traffic = float(digest["hbm_bytes_per_launch"]) if (digest and batch == cfg_batch) else None
        out = {
            "metric": metric,
            "value": samples / t_all / 1e9,
            "unit": "Gsample/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_all / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": line_dtype,
            "data": "synthetic",
            "config": {"workload": "%s, batch=%d per GPU" % (workload, batch),
                       "batch_per_gpu": batch, "n": n, "parallelism": "batch-shard x%d" % world,
                       "kernel": core.info["kernel_name"], "launches_per_step": core.info["n_passes"],
                       "clock_prewarm_steps": args.prewarm,
                       "backend": backend, "rccl_ranks": rccl_ranks,
                       "data_per_rank": "seed + rank" if args.rank_seeded_data else "identical on every rank (rank 0's batch)"},
            "per_gpu": {"kernel_ms": per_gpu_ms,
                        "Gsample/s": [batch * n / m / 1e6 for m in per_gpu_ms],
                        "roofline_frac": [alg_bytes / (m * 1e-3) / 1e9 / HBM_PEAK_GBS for m in per_gpu_ms]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": ("%s (static: rocprofv3 --pmc passes of this command, committed; not measured in this run)"
                                            % digest_path) if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": kern_ms},
            "lib_version": loaded_lib_version(),
        }
        if traffic:  # the digest names the sources it was taken on: a kernel change that was not re-profiled shows here
            out["roofline"]["traffic_stale"] = digest_stale(digest)
            out["roofline"]["traffic_lib_version"] = digest.get("lib_version")
        if e2e:
            out["e2e"] = e2e
        elif args.e2e:
            out["e2e"] = None
            out["e2e_note"] = "--e2e covers the int16 -> int16 configurations (C2, C5); not measured for %s" % args.config
        if extras:
            out["cold"] = cold
            xf = make_input(batch, n, seed, 0, full_scale=True, dw=dw)
            stepf = lambda: core.exec_raw(xf.data_ptr(), out_ptr, batch, stream)  # noqa: E731
            for _ in range(20):
                stepf()
            ms = event_ms(torch, stepf, args.steps)
            out["full_scale_input"] = {"kernel_ms": ms, "value": batch * n / ms / 1e6,
                                       "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "what": "input uniform over the whole DATA_WIDTH range: no frame has a guard bit, every frame takes "
                                               "the exact result extraction (the headline input is uniform in [-2^14, 2^14) as SURVEY 8d names)"}
            del xf
            step()  # y := transform(x) again for the parity gate below
            torch.cuda.synchronize()
            # the copy ceiling is the single-pass kernels' access pattern (one wave per 4 KiB): C2 / C5; the VALU-issue bound: every config
            same_batch = bool(digest) and digest.get("digest_batch") in (None, batch)
            out.update(ceilings(torch, x, torch.empty_like(x) if args.config in ("C2", "C5") else None, stream, digest if same_batch else None,
                                args.config == "C2" and batch == 65536, samples_per_call=float(batch) * n, copy=args.config in ("C2", "C5")))
            vb = out.get("valu_bound", {})
            if vb.get("value"):  # name the roofline that binds this configuration: whichever of HBM bytes / VALU issue allows less
                hbm_bound = HBM_PEAK_GBS / bytes_per_sample  # Gsample/s at 100 % of the HBM roofline
                here = batch * n / kern_ms / 1e6
                r = out["roofline"]
                r["frac_hbm"] = r["frac"]
                r["frac_valu"] = here / vb["value"]
                bounds = {"hbm": hbm_bound, "valu": vb["value"]}
                if r.get("traffic"):  # multi-pass plans: the bytes their passes really move (PMC digest) against the same 8 TB/s
                    bounds["hbm_pass_traffic"] = HBM_PEAK_GBS / (r["traffic"] / (float(batch) * n))
                    r["frac_hbm_pass_traffic"] = here / bounds["hbm_pass_traffic"]
                r["bound"] = "valu" if vb["value"] < min(v for k, v in bounds.items() if k != "valu") else "hbm"
                r["bounds_Gsample_per_s"] = bounds
                r["note"] = ("`frac`, `achieved`, `peak` are the HBM figures (algorithmic bytes / kernel time over 8 TB/s) for every configuration; "
                             "`bound` names the lowest ceiling: `hbm` (algorithmic bytes, or `hbm_pass_traffic` = the measured bytes of all passes, "
                             "at 8 TB/s) or `valu` (the plan's VALU wave-instructions at the slow-class issue rate measured in this run); "
                             "`frac_valu` / `frac_hbm_pass_traffic` = measured rate over those bounds")
            out["octave"] = octave_probe()
        if world == 1 and not args.no_cpu_baseline:
            step()
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline(x, y, log2n, direction, dw, tw, fmt)
        if world == 1 and args.config == "C2" and not args.batch and not args.no_extras:
            # the VALU-issue floor of the headline itself from the butterfly bodies timed in this run (next to the PMC-based valu_bound)
            rates = body_rates(torch, stream)
            if rates:
                r = out["roofline"]
                here = batch * n / kern_ms / 1e6
                fl = valu_floor(rates, "FWD", 0, False)
                if fl:
                    r["valu_floor_bodies"] = {"value": fl, "unit": "Gsample/s", "frac": here / fl,
                                              "what": "rate VALU issue alone allows: 1.5 wave-rounds of four general stages + 1 of stages 3..0 "
                                                      "per frame, both timed on registers in this run (tools/diag_kernels.hip diag_body)"}
                if "full_scale_input" in out:
                    fl = valu_floor(rates, "FWD", 0, True)
                    if fl:
                        out["full_scale_input"]["valu_floor_bodies"] = {"value": fl, "unit": "Gsample/s",
                                                                         "frac": out["full_scale_input"]["value"] / fl}
                out["body_rates_wave_rounds_per_s"] = {"kind%d_round%d_fastx%d" % k: v for k, v in sorted(rates.items())}
            if not args.no_other_modes:
                om = {}
                for name in MODES:
                    try:
                        om[name] = other_mode(torch, name, args.steps, args.warmup, dev_index, rates, cpu=not args.no_cpu_baseline)
                    except Exception as exc:  # a sub-record must never cost the headline line
                        om[name] = {"error": repr(exc)}
                        torch.cuda.empty_cache()
                out["other_modes"] = om
        if world == 1 and args.config == "C2" and not args.batch and not args.no_other_configs and not args.no_extras:
            # BASELINE's other configurations, timed in this process after everything that belongs to the headline (the headline fields
            # above are complete and unchanged); LAST key of the line, so that a tail of it shows the three sub-records
            slow = out.get("valu_bound", {}).get("slow_class_wave_insts_per_s")
            del x, y
            torch.cuda.empty_cache()
            oc = {}
            for name in ("C3", "C4", "C5"):
                try:
                    oc[name] = other_config(torch, name, args.steps, args.warmup, dev_index, slow, cpu=not args.no_cpu_baseline)
                except Exception as exc:  # a sub-record must never cost the headline line
                    oc[name] = {"error": repr(exc)}
                    torch.cuda.empty_cache()
            out["other_configs"] = oc
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
