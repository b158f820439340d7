"""tools/pmc_digest.py <dir made by tools/pmc_digest.sh> <spec> -> JSON digest on stdout (per kernel: mean counters per launch, durations,
derived HBM bytes; per config: measured traffic of ONE intfft_exec call over its algorithmic bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, spec = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as BC  # noqa: E402  (CONFIGS / adhoc only; nothing runs)

if spec not in BC.CONFIGS:
    BC.adhoc(spec)
log2n, dw, tw, fmt, rnd, direction, batch, bits, bps = BC.CONFIGS[spec]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("intfft::", "")


kern = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if "k_" in n and "twiddle" not in n and "pack" not in n:
            kern[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if "k_" in n and "twiddle" not in n and "pack" not in n:
            dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
try:  # which library the counters were taken on: intfft_version() carries the hash of its sources (intfftk_amd/build.py source_hash)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from intfftk_amd import _capi as _capi  # noqa: E402

    lib_version = _capi.lib().intfft_version().decode()
except Exception as exc:  # noqa: BLE001
    lib_version = "unknown (%r)" % (exc,)
out = {"config": spec, "log2n": log2n, "batch": batch, "direction": direction, "bytes_per_sample": bps, "kernels": {}, "lib_version": lib_version}
alg = bps * (1 << log2n) * batch
out["algorithmic_bytes_per_call"] = alg
# launches per intfft_exec call: the trace holds (warm-up + timed) calls; every kernel of the plan is launched the same number of times
# per chunk, so bytes per call = sum over kernels of (bytes per launch x launches) / calls, calls = launches of the rarest kernel / its
# launches per call -- simpler and robust: total bytes of the whole run / total calls, with calls taken from the bench's own step count.
total_bytes = 0.0
for n, c in kern.items():
    d = {k: sum(v) / len(v) for k, v in c.items()}
    d["launches"] = len(dur.get(n, [])) or len(next(iter(c.values())))
    if dur.get(n):
        tail = dur[n][-max(1, len(dur[n]) // 4):]  # the last quarter of the launches: clocks ramped
        d["avg_ns"] = sum(dur[n]) / len(dur[n])
        d["steady_avg_ns"] = sum(tail) / len(tail)
    else:
        d["avg_ns"] = d["steady_avg_ns"] = 0.0
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2  # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_EA0_RDREQ_sum" in d:
        d["crosscheck_rdreq_x128B"] = d["TCC_EA0_RDREQ_sum"] * 128
        d["crosscheck_wrreq_x64B"] = d.get("TCC_EA0_WRREQ_sum", 0) * 64
    if "TCC_EA0_RDREQ_128B_sum" in d:  # round 4: bytes by request size (tools/reqbench.hip: every read miss is a 128-byte request,
        # also for 64- / 32-byte pieces; writes leave as 64-byte requests, 32-byte ones for 32-byte pieces)
        d["rdreq_bytes_by_size"] = d["TCC_EA0_RDREQ_128B_sum"] * 128 + d.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + d.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
        if "TCC_EA0_WRREQ_sum" in d and "TCC_EA0_WRREQ_64B_sum" in d:
            d["wrreq_bytes_by_size"] = d["TCC_EA0_WRREQ_64B_sum"] * 64 + (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"]) * 32
    out["kernels"][n] = d
# launches per intfft_exec call from the trace run (bench_configs.py prints its number of calls)
calls = None
try:
    for line in open(os.path.join(root, "trace.log")):
        if line.startswith("{"):
            calls = json.loads(line).get("calls")
except OSError:
    pass
out["calls_in_trace_run"] = calls
if calls:
    per_call = 0.0
    for n, d in out["kernels"].items():
        d["launches_per_call"] = len(dur.get(n, [])) / calls
        per_call += (d.get("hbm_read_bytes", 0.0) + d.get("hbm_write_bytes", 0.0)) * d["launches_per_call"]
    out["hbm_bytes_per_call"] = per_call
    out["traffic_over_algorithmic"] = per_call / alg
else:
    out["traffic_over_algorithmic"] = None
print(json.dumps(out, indent=1, sort_keys=True))
