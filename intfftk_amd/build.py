"""Builds intfftk_amd/lib/libintfft.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libintfft.so")
SOURCES = ["intfft_plan.hip", "intfft_generic.hip", "intfft_pass16.hip", "intfft_fastsmall.hip", "intfft_fast1024.hip", "intfft_fast1024x.hip", "intfft_fast1024u.hip", "intfft_fast1024ux.hip", "intfft_fastw32.hip", "intfft_fastw64.hip", "intfft_fastw64s.hip", "intfft_fastw64n.hip", "intfft_fastw64sn.hip", "intfft_fastw64bn.hip", "intfft_fastw64b.hip", "intfft_fastw64bi.hip", "intfft_fast4096w.hip", "intfft_w32inv.hip", "intfft_bigw.hip", "intfft_fast4096.hip", "intfft_fast16k.hip", "intfft_big20.hip", "intfft_big2p.hip", "intfft_big2x.hip", "intfft_wide16.hip", "intfft_widelong.hip", "intfft_bigwlong.hip", "intfft_reorder.hip", "intfft_stream.hip"]
HEADERS = ["intfft_device.hpp", "intfft_internal.hpp", "intfft_pk16.hpp", "intfft_u32.hpp", "intfft_w64.hpp", os.path.join("..", "..", "include", "intfft.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


VERSION_SRC = "intfft_version.hip"


def source_hash() -> str:
    """sha256 (first 16 hex digits) over every file of csrc/ (names and contents, sorted) and include/intfft.h: what intfft_version()
    reports, so that measurements can name the sources of the library they were taken on."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    for f in files + [os.path.join("..", "..", "include", "intfft.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    # the version unit is rebuilt whenever the hash of the sources changes (the hash it was built with is kept beside the object)
    sh = source_hash()
    vo, vstamp = os.path.join(LIBDIR, "intfft_version.o"), os.path.join(LIBDIR, "intfft_version.hash")
    try:
        with open(vstamp) as fh:
            have = fh.read().strip()
    except OSError:
        have = ""
    if force or have != sh or not os.path.exists(vo):
        jobs.append([HIPCC] + FLAGS + ['-DINTFFT_SRC_HASH="%s"' % sh, "-c", os.path.join(CSRC, VERSION_SRC), "-o", vo])
    objs.append(vo)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        extra = [os.path.join(CSRC, "intfft_wide16.hip")] if src == "intfft_widelong.hip" else []  # includes its kernel templates
        if force or _stale(o, [s] + hdrs + extra):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])
        objs.append(o)
    if jobs:  # independent translation units: compile them side by side, the long ones first (measured seconds on one core; unlisted: 10)
        from concurrent.futures import ThreadPoolExecutor

        cost = {"intfft_bigw": 58, "intfft_w32inv": 38, "intfft_fastw64n": 34, "intfft_fastw64": 34, "intfft_fastw64bn": 34, "intfft_wide16": 29, "intfft_big2x": 27, "intfft_widelong": 23,
                "intfft_fastw32": 26, "intfft_fastw64s": 21, "intfft_fast1024": 21, "intfft_fastw64sn": 21, "intfft_generic": 20, "intfft_fast16k": 20,
                "intfft_fastw64bi": 19, "intfft_fast4096w": 19, "intfft_fastw64b": 16, "intfft_big20": 15, "intfft_fast1024x": 13, "intfft_fast1024ux": 12,
                "intfft_fast4096": 12}
        jobs.sort(key=lambda cmd: -cost.get(os.path.basename(cmd[-3]).replace(".hip", ""), 10))

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        workers = max(1, min(len(jobs), int(os.environ.get("INTFFT_BUILD_JOBS", os.cpu_count() or 1))))
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(run, jobs))
        with open(vstamp, "w") as fh:
            fh.write(sh + "\n")
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


DIAG_SRC = os.path.join(HERE, "..", "tools", "diag_kernels.hip")
DIAG_LIB = os.path.join(HERE, "..", "tools", "lib", "libintfft_diag.so")


def build_diag(force: bool = False, verbose: bool = False) -> str:
    """tools/lib/libintfft_diag.so: the on-box ceilings bench.py prints next to its line (copy ceiling of the headline
    kernel's access pattern, VALU issue rate).  Diagnostics only -- not linked into, or loaded by, libintfft.so."""
    os.makedirs(os.path.dirname(DIAG_LIB), exist_ok=True)
    if force or _stale(DIAG_LIB, [DIAG_SRC]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result", "-o", DIAG_LIB, DIAG_SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return DIAG_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_diag(force="--force" in sys.argv, verbose=True))
