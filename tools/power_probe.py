"""tools/power_probe.py [rnd] [direction]: is the rate of ONE instruction stream data dependent?  Runs the N = 1024 16-bit scaled plan
(default: RNDMODE = 1 forward -- a kernel with NO data-dependent branch: no fast / exact alternative, no vote) on four input
distributions, 400 launches each after a clock ramp, and prints the HIP-event rate per phase.  Under
`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` (tools/power_probe.sh) the per-dispatch GRBM_GUI_ACTIVE / 8 XCDs / duration is the shader
clock the phase ran at: a lower clock on high-entropy data at equal instruction counts = the power limit, not the kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from intfftk_amd import IntFFTCore

rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
direction = sys.argv[2] if len(sys.argv) > 2 else "FWD"
N, B = 1024, 65536
core = IntFFTCore(10, 16, 16, 0, rnd, "NEW", direction)
g = torch.Generator(device="cuda")
g.manual_seed(7)
inputs = {
    "zeros": torch.zeros((B, N, 2), dtype=torch.int16, device="cuda"),
    "uniform_13bit": torch.randint(-(1 << 12), 1 << 12, (B, N, 2), dtype=torch.int16, device="cuda", generator=g),
    "uniform_15bit": torch.randint(-(1 << 14), 1 << 14, (B, N, 2), dtype=torch.int16, device="cuda", generator=g),
    "uniform_16bit": torch.randint(-(1 << 15), 1 << 15, (B, N, 2), dtype=torch.int16, device="cuda", generator=g),
}
y = torch.empty((B, N, 2), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("PROBE_REPS", "400"))
for _ in range(reps):  # ramp
    core.exec_raw(inputs["uniform_15bit"].data_ptr(), y.data_ptr(), B, st)
torch.cuda.synchronize()
for name, x in inputs.items():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(50):
        core.exec_raw(x.data_ptr(), y.data_ptr(), B, st)
    e0.record()
    for _ in range(reps):
        core.exec_raw(x.data_ptr(), y.data_ptr(), B, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"input": name, "rnd": rnd, "dir": direction, "kernel": core.info["kernel_name"], "ms": ms,
                      "Gsample/s": B * N / ms / 1e6, "launches": reps + 50}), flush=True)
