"""GPU tests of the C-ABI surface added in round 2: the C driver of INTEGRATION.md (no Python in the loop), the
standalone re-orderer, the overlap rule of intfft_exec, intfft_shard_prepare, the RCCL ("nccl") backend under
ShardedTransform, and bench.py's own N-rank launch."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle_c as C
from tests.helpers import edge_frames, uniform_frames

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NP_DT = {2: np.int16, 4: np.int32, 8: np.int64}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---- the C driver: compiled here with gcc against include/intfft.h, run as its own process ------------------------
@pytest.fixture(scope="module")
def c_driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cdrv") / "driver")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I%s/include" % rocm,
           "-I%s" % os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_driver", "driver.c"), "-o", exe,
           "-L%s" % os.path.join(ROOT, "intfftk_amd", "lib"), "-lintfft", "-L%s/lib" % rocm, "-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.parametrize("nfft,mode,batch", [(7, 0, 33), (7, 1, 33), (7, 2, 33), (10, 0, 100), (10, 2, 9), (13, 0, 3)])
def test_c_driver_matches_oracle(c_driver, tmp_path, nfft, mode, batch):
    """fft_signle_test's three UUTs (TRUNCATE / ROUNDING / UNSCALED, fft_signle_test.vhd:93-112) through the plain-C binding."""
    n = 1 << nfft
    x = np.concatenate([edge_frames(n, 16), uniform_frames(max(batch - 8, 1), n, 16, 4242 + nfft + mode)])[-batch:]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    x.astype(np.int16).tofile(fin)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "intfftk_amd", "lib"), "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([c_driver, fin, fout, str(batch), str(nfft), str(mode)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    fmt, rnd = mode // 2, mode % 2
    p = C.make_params(nfft, 16, 16, fmt, rnd, True)
    want = C.execute(x, p, C.FWD)
    cb = 2 if not fmt else (4 if 16 + nfft <= 32 else 8)
    got = np.fromfile(fout, dtype=NP_DT[cb]).reshape(batch, n, 2).astype(np.int64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_c_driver_sharded(c_driver, tmp_path, transport):
    """driver --sharded: one plan per visible device, intfft_shard_prepare + intfft_shard_set_transport + intfft_exec_sharded from plain C
    (RCCL through the library's dlopen -- the driver itself links neither RCCL nor torch); the result is the oracle's."""
    nfft, batch = 10, 77
    n = 1 << nfft
    x = uniform_frames(batch, n, 16, 5151)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    x.astype(np.int16).tofile(fin)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "intfftk_amd", "lib"), "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [c_driver, "--sharded", transport, fin, fout, str(batch), str(nfft), "0"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        # seen once in round 6 on one box (a run that normally takes 3 s; six repetitions on the next box: 6 x 3 s): the communicator's
        # start-up did not return.  One more try; a second hang fails the test.
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert ("transport rccl" if transport == "rccl" else "transport peer copies") in r.stdout, r.stdout
    got = np.fromfile(fout, dtype=np.int16).reshape(batch, n, 2).astype(np.int64)
    assert np.array_equal(got, C.execute(x, C.make_params(nfft, 16, 16, 0, 0, True), C.FWD))


# ---- intfft_reorder: the buffers/ blocks as an operator ------------------------------------------------------------
def _order_index(order, log2n):
    L = C.lib()
    return np.array([L.orc_order_index(order, log2n, m) for m in range(1 << log2n)], dtype=np.int64)


@pytest.mark.parametrize("log2n", [3, 5, 6, 7, 10, 12, 13, 16])
@pytest.mark.parametrize("cb", [2, 4, 8])
def test_reorder_all_order_pairs(log2n, cb):
    """d_out[m_out] = d_in[m_in] with the same logical index on both sides, for all 16 (from, to) pairs -- among them
    int_bitrev_order (BITREV_LANES -> NATURAL, int_bitrev_order.vhd:82-104) and bitrevorder (NATURAL <-> BITREV)."""
    import torch

    from intfftk_amd import _capi as capi

    n = 1 << log2n
    batch = 3 if log2n >= 12 else 7
    rng = np.random.default_rng(log2n * 10 + cb)
    x = rng.integers(-(1 << 14), 1 << 14, size=(batch, n, 2)).astype(NP_DT[cb])
    xd = torch.from_numpy(x).cuda()
    maps = {o: _order_index(o, log2n) for o in range(4)}  # memory index -> logical index
    for fo in range(4):
        inv_from = np.empty(n, dtype=np.int64)
        inv_from[maps[fo]] = np.arange(n)  # logical -> memory index of the input
        for to in range(4):
            yd = torch.zeros_like(xd)
            rc = capi.lib().intfft_reorder(log2n, cb, fo, to, xd.data_ptr(), yd.data_ptr(), batch, 0, None)
            assert rc == 0, capi.strerror(rc)
            torch.cuda.synchronize()
            want = x[:, inv_from[maps[to]], :]
            assert np.array_equal(yd.cpu().numpy(), want), (fo, to)


def test_reorder_then_transform_equals_ordered_plan():
    """The wrappers' composition (int_fft_single_path.vhd:157-268): int_fftNk in its native orders, followed by the
    BITREV -> NATURAL re-order, equals the NATURAL-order plan; errors are status codes."""
    import torch

    from intfftk_amd import _capi as capi
    from intfftk_amd import int_fft_single_path, int_fftNk

    x = uniform_frames(9, 1024, 15, 99).astype(np.int16)
    xd = torch.from_numpy(x).cuda()
    halves = torch.empty_like(xd)
    L = capi.lib()
    assert L.intfft_reorder(10, 2, capi.ORDER_NATURAL, capi.ORDER_HALVES, xd.data_ptr(), halves.data_ptr(), 9, 0, None) == 0
    beats = int_fftNk(10, 16, 16, 0, 0)(halves)
    nat = torch.empty_like(beats)
    assert L.intfft_reorder(10, 2, capi.ORDER_BITREV, capi.ORDER_NATURAL, beats.data_ptr(), nat.data_ptr(), 9, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(nat, int_fft_single_path(10, 16, 16, 0, 0)(xd))
    assert L.intfft_reorder(10, 2, 0, 1, xd.data_ptr(), xd.data_ptr(), 9, 0, None) == capi.ERR_INVALID  # in place
    assert L.intfft_reorder(10, 3, 0, 1, xd.data_ptr(), nat.data_ptr(), 9, 0, None) == capi.ERR_INVALID
    assert L.intfft_reorder(2, 2, 0, 1, xd.data_ptr(), nat.data_ptr(), 9, 0, None) == capi.ERR_INVALID
    assert L.intfft_reorder(10, 2, 0, 4, xd.data_ptr(), nat.data_ptr(), 9, 0, None) == capi.ERR_INVALID
    assert L.intfft_reorder(10, 2, 0, 1, None, nat.data_ptr(), 9, 0, None) == capi.ERR_NULL
    assert L.intfft_reorder(10, 2, 0, 1, xd.data_ptr(), nat.data_ptr(), 9, 99, None) == capi.ERR_NO_DEVICE


# ---- intfft_exec: in place only as d_in == d_out with equal containers ---------------------------------------------
def test_exec_rejects_partial_overlap():
    import torch

    from intfftk_amd import IntFFTCore
    from intfftk_amd import _capi as capi

    scaled = IntFFTCore(10, 16, 16, 0, 0)
    x = torch.from_numpy(uniform_frames(8, 1024, 15, 5).astype(np.int16)).cuda()
    want = scaled(x).clone()
    L = capi.lib()
    buf = torch.zeros((9, 1024, 2), dtype=torch.int16, device="cuda")
    buf[:8] = x
    assert L.intfft_exec(scaled._plan, buf.data_ptr(), buf.data_ptr() + 4096, 8, None) == capi.ERR_INVALID  # shifted by a frame
    assert L.intfft_exec(scaled._plan, buf.data_ptr(), buf.data_ptr(), 8, None) == 0                          # in place
    torch.cuda.synchronize()
    assert torch.equal(buf[:8], want)
    unscaled = IntFFTCore(10, 16, 16, 1, 0)  # int16 in, int32 out: in place would overwrite unread frames
    big = torch.zeros((8, 1024, 2), dtype=torch.int32, device="cuda")
    assert L.intfft_exec(unscaled._plan, big.data_ptr(), big.data_ptr(), 8, None) == capi.ERR_INVALID
    assert L.intfft_exec(unscaled._plan, big.data_ptr() + 2 * 1024 * 2 * 4 - 2, big.data_ptr(), 2, None) == capi.ERR_INVALID  # last input bytes inside the output range


def test_io_widths_and_plan_create_agree():
    """validate() is shared: whatever intfft_io_widths accepts, intfft_plan_create elaborates (and the oracle agrees)."""
    from intfftk_amd import _capi as capi

    L = capi.lib()
    for (dw, tw, xser, fmt) in [(16, 26, 0, 0), (16, 25, 0, 0), (16, 28, 1, 0), (16, 27, 1, 0), (30, 5, 0, 0), (30, 6, 0, 0),
                                (30, 5, 1, 0), (50, 16, 1, 1), (60, 16, 1, 0), (40, 24, 1, 0)]:
        p = capi.Params(6, dw, tw, fmt, 0, xser, capi.FWD, 1, 0, 0)
        a = L.intfft_io_widths(ctypes.byref(p), None, None, None, None)
        plan = ctypes.c_void_p()
        b = L.intfft_plan_create(ctypes.byref(plan), ctypes.byref(p), 0)
        if plan.value:
            L.intfft_plan_destroy(plan)
        oracle_ok = C.lib().orc_validate(ctypes.byref(C.make_params(6, dw, tw, fmt, 0, bool(xser))), C.FWD) == 0
        assert a == b, (dw, tw, xser, a, b)
        assert (a == 0) == oracle_ok, (dw, tw, xser, a, oracle_ok)


# ---- intfft_shard_prepare ------------------------------------------------------------------------------------------
def test_shard_prepare_then_exec_sharded():
    import torch

    from intfftk_amd import IntFFTCore, exec_sharded
    from intfftk_amd import _capi as capi

    ndev = torch.cuda.device_count()
    cores = [IntFFTCore(12, 16, 16, 0, 0, "NEW", "PAIR", device=i % ndev) for i in range(3)]
    arr = (ctypes.c_void_p * 3)(*[c._plan for c in cores])
    assert capi.lib().intfft_shard_prepare(arr, 3, 1, 64) == 0
    x = uniform_frames(50, 4096, 15, 8)
    s = torch.cuda.Stream()  # produced on a side stream: the call waits for the whole root device (contract in intfft.h)
    with torch.cuda.stream(s):
        xd = torch.from_numpy(x.astype(np.int16)).to("cuda:%d" % (1 % ndev), non_blocking=True)
    y = exec_sharded(cores, xd, 1)
    want = C.execute(x, C.make_params(12, 16, 16, 0, 0, True), C.PAIR)
    assert np.array_equal(y.cpu().numpy().astype(np.int64), want)
    assert capi.lib().intfft_shard_prepare(arr, 3, 7, 64) == capi.ERR_INVALID
    with pytest.raises(ValueError):
        exec_sharded(cores, xd.cpu(), 1)
    with pytest.raises(ValueError):
        exec_sharded([cores[0], cores[0]], xd, 0)
    for c in cores:
        c.close()


def test_exec_sharded_rccl_transport_equals_peer_copies():
    """intfft_shard_set_transport(INTFFT_TRANSPORT_RCCL): the scatter and the gather of intfft_exec_sharded as ONE group of ncclSend / ncclRecv
    each (librccl.so through dlopen, communicators owned by the plan set), one plan per visible device (world = device_count: 1 on the
    single-GPU box -- the root transforms in place and both groups are empty --, 8 on a full node), bit for bit the result of the peer-copy
    transport and of the oracle; two plans on one device are refused by RCCL and the set stays on peer copies; odd batches (remainder to the
    last plans), a batch smaller than the set."""
    import torch

    from intfftk_amd import IntFFTCore, exec_sharded
    from intfftk_amd import _capi as capi

    ndev = torch.cuda.device_count()
    cores = [IntFFTCore(10, 16, 16, 0, 0, "NEW", "FWD", device=i) for i in range(ndev)]
    p = C.make_params(10, 16, 16, 0, 0, True)
    for batch in (max(1, ndev - 1), 5 * ndev + 3, 4096 + 1):
        x = uniform_frames(batch, 1024, 15, 90 + batch)
        xd = torch.from_numpy(x.astype(np.int16)).to("cuda:0")
        y_rccl = exec_sharded(cores, xd, 0, transport="rccl")
        y_peer = exec_sharded(cores, xd, 0, transport="peer")
        assert torch.equal(y_rccl, y_peer)
        sel = sorted({0, batch // 2, batch - 1})
        assert np.array_equal(y_rccl[sel].cpu().numpy().astype(np.int64), C.execute(x[sel], p, C.FWD))
    if ndev > 1:  # another root
        xd1 = xd.to("cuda:1")
        assert torch.equal(exec_sharded(cores, xd1, 1, transport="rccl").cpu(), y_peer.cpu())
    # one rank per device: a set with two plans on device 0 cannot have communicators; it keeps working on peer copies
    twin = IntFFTCore(10, 16, 16, 0, 0, "NEW", "FWD", device=0)
    arr = (ctypes.c_void_p * 2)(cores[0]._plan, twin._plan)
    assert capi.lib().intfft_shard_set_transport(arr, 2, 0, capi.TRANSPORT_RCCL) == capi.ERR_TRANSPORT
    assert "transport" in capi.strerror(capi.ERR_TRANSPORT)
    assert capi.lib().intfft_shard_set_transport(arr, 2, 0, 7) == capi.ERR_INVALID
    assert torch.equal(exec_sharded([cores[0], twin], xd, 0), y_peer)
    twin.close()
    for c in cores:
        c.close()


def test_exec_host_rejects_wrong_dtype():
    from intfftk_amd import IntFFTCore

    core = IntFFTCore(7, 16, 16, 0, 0)
    with pytest.raises(TypeError):
        core.exec_host(np.zeros((2, 128, 2), dtype=np.int32))
    with pytest.raises(TypeError):
        core.exec_host(np.zeros((2, 128, 2), dtype=np.float64))
    assert core.exec_host(np.zeros((2, 128, 2), dtype=np.int16)).shape == (2, 128, 2)


# ---- RCCL under ShardedTransform: world = every visible GPU --------------------------------------------------------
_NCCL_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from intfftk_amd import int_fft_single_path
from intfftk_amd.sharding import ShardedTransform, max_over_ranks, gather_floats
core = int_fft_single_path(NFFT=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=0, RNDMODE=0, device=rank)
sh = ShardedTransform(core, 1024, core.in_dtype, core.out_dtype, dev)
batch = 37
g = torch.Generator(device="cpu"); g.manual_seed(7)
full = torch.randint(-2 ** 14, 2 ** 14, (batch, 1024, 2), dtype=torch.int16, generator=g)
out = sh.run_from_root(full.to(dev) if rank == 0 else None, batch, 0)
t = torch.ones(1, device=dev); dist.all_reduce(t)
assert int(t.item()) == world
assert abs(max_over_ranks(1.0 + rank, dev) - world) < 1e-12
assert gather_floats(float(rank), dev) == [float(r) for r in range(world)]
if rank == 0:
    torch.save(out.cpu(), os.environ["OUT_FILE"])
    torch.save(full, os.environ["IN_FILE"])
dist.barrier(); dist.destroy_process_group()
"""


def test_sharded_transform_on_rccl(tmp_path):
    """scatter -> transform -> gather with backend "nccl" (= RCCL) at world = device_count: grouped ncclSend / ncclRecv
    between GPUs where the box has several, the degenerate one-rank group on a 1-GPU box (same code path)."""
    import torch

    world = torch.cuda.device_count()
    port = _free_port()
    script = tmp_path / "w.py"
    script.write_text(_NCCL_WORKER % {"root": ROOT})
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OUT_FILE=str(tmp_path / "out.pt"), IN_FILE=str(tmp_path / "in.pt"), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    full = torch.load(tmp_path / "in.pt").numpy()
    out = torch.load(tmp_path / "out.pt").numpy()
    want = C.execute_i16(full, C.make_params(10, 16, 16, 0, 0, True), C.FWD)
    assert np.array_equal(out, want)


# ---- bench.py launches its own ranks -------------------------------------------------------------------------------
def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_spawns_n_ranks():
    """`python bench.py --gpus 2` with no launcher: two ranks, ONE line, n_gpus = 2.  With fewer than 2 GPUs the ranks
    share the device over gloo (INTFFT_BENCH_SHARE_GPU=1, diagnostics); with >= 2 GPUs this runs over RCCL."""
    import torch

    out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm", "5", "--batch", "2048", "--e2e", "--no-cpu-baseline"],
                 {"INTFFT_BENCH_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2
    assert out["config"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert len(out["per_gpu"]["kernel_ms"]) == 2 and all(v > 0 for v in out["per_gpu"]["Gsample/s"])
    assert out["e2e"]["matches_resident"] is True and out["e2e"]["frames"] == 4096
    assert out["e2e"]["p2p_ops_per_group"] == [1, 1]  # the root's scatter group and gather group: one peer each
    assert out["e2e"]["pipelined"]["matches_serial"] is True and out["e2e"]["pipelined"]["pieces"] == 4 and out["e2e"]["pipelined"]["value"] > 0
    assert out["scaling"] == "weak" and out["unit"] == "Gsample/s"


def test_bench_single_gpu_line_has_the_8d_fields():
    out = _bench(["--steps", "5", "--warmup", "2", "--prewarm", "20", "--batch", "4096", "--e2e"])
    assert out["n_gpus"] == 1 and out["metric"].startswith("Gsample/s (complex int16) batched N=1024")
    for key in ("roofline", "cpu_baseline", "cold", "full_scale_input", "copy_ceiling", "valu_bound", "octave", "e2e", "per_gpu"):
        assert key in out, key
    assert out["cpu_baseline"]["parity_ok"] is True and out["cpu_baseline"]["in_place_form"]["parity_ok"] is True
    assert out["cpu_baseline"]["single_thread"]["cores"] == 1
    assert out["roofline"]["traffic"] is None  # not BASELINE's batch: no static PMC figure is attached
    assert out["e2e"]["matches_resident"] is True
    c5 = _bench(["--config", "C5", "--steps", "3", "--warmup", "1", "--prewarm", "5", "--batch", "512", "--no-extras"])
    assert c5["config"]["n"] == 4096 and c5["cpu_baseline"]["parity_ok"] is True
    # the other BASELINE configurations in the same line format (reduced batches: their CPU samples are a few frames)
    c3 = _bench(["--config", "C3", "--steps", "3", "--warmup", "1", "--prewarm", "5", "--batch", "16", "--no-extras"])
    assert c3["config"]["n"] == 65536 and c3["dtype"] == "int64" and c3["config"]["launches_per_step"] == 2 and c3["cpu_baseline"]["parity_ok"] is True
    c4 = _bench(["--config", "C4", "--steps", "3", "--warmup", "1", "--prewarm", "5", "--batch", "8", "--no-extras"])
    assert c4["config"]["n"] == 1 << 20 and c4["config"]["kernel"] == "k_big2x_a/k_big2x_b" and c4["cpu_baseline"]["parity_ok"] is True


def test_bench_default_line_carries_other_configs():
    """The driver's single-GPU command (BASELINE's batches, no --batch): the headline fields as before, and as the LAST key the
    `other_configs` sub-records -- C3, C4, C5 timed in the same process, each with value / ms_per_step / kernel_ms / roofline
    {bound, frac_hbm, frac_valu, frac_hbm_pass_traffic, traffic} and the oracle on a prefix as parity gate."""
    out = _bench(["--steps", "5", "--warmup", "2", "--prewarm", "50"])
    assert out["n_gpus"] == 1 and out["config"]["batch_per_gpu"] == 65536 and out["cpu_baseline"]["parity_ok"] is True
    assert list(out)[-1] == "other_configs" and sorted(out["other_configs"]) == ["C3", "C4", "C5"]
    want_n = {"C3": 65536, "C4": 1 << 20, "C5": 4096}
    for name, rec in out["other_configs"].items():
        assert "error" not in rec, (name, rec)
        assert ("N=%d" % want_n[name] in rec["workload"]) or ("N=2^20" in rec["workload"])
        assert rec["steps"] == 5 and rec["value"] > 0 and rec["kernel_ms"] > 0 and rec["ms_per_step"] >= 0.9 * rec["kernel_ms"]
        r = rec["roofline"]
        for key in ("bound", "frac", "frac_hbm", "frac_valu", "frac_hbm_pass_traffic", "traffic"):
            assert key in r, (name, key)
        assert r["bound"] in ("hbm", "valu") and 0 < r["frac"] < 1 and r["traffic"] >= r["algorithmic_bytes_per_step"]
        assert rec["cpu_baseline"]["parity_ok"] is True and rec["cpu_baseline"]["parity_checked_frames"] >= 8
    assert out["other_configs"]["C4"]["kernel"] == "k_big2x_a/k_big2x_b" and out["other_configs"]["C3"]["launches_per_step"] == 2


def test_bench_under_the_drivers_launcher():
    """The driver's form for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE from the environment, ONE JSON line from rank 0."""
    import torch

    env = dict(os.environ, INTFFT_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--prewarm", "5", "--batch", "1024"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and "cpu_baseline" not in out
    assert out["config"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert len(out["per_gpu"]["Gsample/s"]) == 2 and out["value"] > 0
    if torch.cuda.device_count() >= 2:  # ranks on their own GPUs: the aggregate is about the sum of the ranks (on one shared GPU they contend)
        assert abs(out["value"] - sum(out["per_gpu"]["Gsample/s"])) / out["value"] < 0.5


def test_chunked_plans_on_two_streams_equal_one_stream(monkeypatch):
    """The two-pass plans of N = 2^19 / 2^20 and the 24-bit-class plans run the scratch-sized chunks of a large batch alternately on
    the caller's stream and a plan-owned side stream (fork / join with events).  Same bits as on one stream (INTFFT_ONE_STREAM), for a
    batch of several chunks, back-to-back calls, a non-default caller stream, and work queued behind the call on that stream."""
    import torch

    from intfftk_amd import IntFFTCore

    monkeypatch.setenv("INTFFT_SCRATCH_MB", "16")  # 4 frames of N = 2^20 (8 of N = 2^16 at 24 bits) per chunk: many chunks in a small batch
    for log2n, dw, tw, fmt, dt, batch in ((20, 16, 16, 0, torch.int16, 19), (16, 24, 24, 1, torch.int32, 37)):
        n = 1 << log2n
        g = torch.Generator(device="cuda")
        g.manual_seed(5 + log2n)
        x = torch.randint(-(1 << (dw - 2)), 1 << (dw - 2), (batch, n, 2), device="cuda", dtype=dt, generator=g)
        core = IntFFTCore(log2n, dw, tw, fmt, 0, "NEW", "FWD", "NATURAL", "NATURAL")
        s = torch.cuda.Stream()
        y = torch.empty(core.out_shape(batch), device="cuda", dtype=core.out_dtype)
        z = torch.empty_like(y)
        with torch.cuda.stream(s):
            core.exec_raw(x.data_ptr(), y.data_ptr(), batch, s.cuda_stream)
            core.exec_raw(x.data_ptr(), z.data_ptr(), batch, s.cuda_stream)  # back to back: the second fork follows the first join
            total = z.to(torch.int64).sum()  # queued behind the call on the caller's stream: must see every chunk's output
        s.synchronize()
        core.close()
        monkeypatch.setenv("INTFFT_ONE_STREAM", "1")
        ref = IntFFTCore(log2n, dw, tw, fmt, 0, "NEW", "FWD", "NATURAL", "NATURAL")
        w = ref(x)
        torch.cuda.synchronize()
        ref.close()
        monkeypatch.delenv("INTFFT_ONE_STREAM")
        assert torch.equal(y, w) and torch.equal(z, w) and int(total) == int(w.to(torch.int64).sum())


@pytest.mark.parametrize("cfg", [(10, 16, 16, 0, "FWD", 20000), (12, 16, 16, 0, "PAIR", 6000), (14, 16, 16, 0, "INV", 1500), (10, 24, 24, 1, "FWD", 8000)])
def test_single_launch_plans_are_reentrant_across_streams(cfg):
    """"exec is re-entrant across streams".  A plan without plan-owned scratch (info.scratch_bytes == 0: every single-launch plan,
    N <= 16384 on the packed kernels and N <= 4096 elsewhere) holds no mutable state, so ONE plan may be executed on several streams at once
    (include/intfft.h); multi-pass plans own their scratch and must not.  Four streams, interleaved calls on different batches, no host sync in between."""
    import torch

    from intfftk_amd import IntFFTCore

    log2n, dw, tw, fmt, direction, batch = cfg
    core = IntFFTCore(log2n, dw, tw, fmt, 0, "NEW", direction, "NATURAL", "NATURAL")
    assert core.info["scratch_bytes"] == 0 and core.info["n_passes"] == 1, core.info
    n = 1 << log2n
    dt = torch.int16 if dw <= 16 else torch.int32
    g = torch.Generator(device="cuda")
    g.manual_seed(500 + log2n)
    xs = [torch.randint(-(1 << (dw - 2)), 1 << (dw - 2), (batch, n, 2), device="cuda", dtype=dt, generator=g) for _ in range(4)]
    want = [core(x) for x in xs]
    ys = [torch.zeros_like(w) for w in want]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    for rep in range(3):
        for i, st in enumerate(streams):
            core.exec_raw(xs[i].data_ptr(), ys[i].data_ptr(), batch, st.cuda_stream)  # raises on a non-zero status
    torch.cuda.synchronize()
    for y, w in zip(ys, want):
        assert torch.equal(y, w)
    core.close()


WS_CASES = [
    # (log2n, dw, tw, fmt, rnd, direction, l1, batch, INTFFT_SCRATCH_MB): every kind of plan that owns scratch
    (16, 16, 16, 0, 0, "FWD", 0, 24, 1),    # two passes, one scratch on the caller's stream: 4 frames per chunk
    (20, 16, 16, 0, 0, "FWD", 0, 9, 8),     # two passes, two scratch halves on two streams: 2 frames per chunk
    (16, 24, 24, 1, 0, "FWD", 0, 21, 4),    # the 24-bit class (int32 -> int64), two streams
    (20, 16, 16, 0, 0, "FWD", 10, 7, 8),    # the tiled 2-D plan: two launches, two streams
    (21, 16, 16, 0, 0, "FWD", 10, 3, 16),   # 2-D, two launches (k_rows2k_tr; one stream)
    (22, 16, 16, 0, 0, "FWD", 10, 3, 32),   # 2-D, three launches with a row sub-plan
    (21, 16, 16, 0, 0, "INV", 10, 3, 16),   # 2-D inverse, two launches (k_rows2k_qtr; round 5)
    (21, 16, 16, 0, 0, "PAIR", 10, 3, 16),  # 2-D pair, four launches through the second layout buffer (round 5)
    (22, 16, 16, 0, 0, "INV", 10, 3, 32),   # 2-D inverse, three launches with a row sub-plan (round 5)
    (22, 16, 16, 0, 0, "FWD", 11, 3, 32),   # 2-D as 2048 x 2048, two launches (round 5)
    (22, 16, 16, 0, 0, "PAIR", 11, 3, 32),  # ... and its pair, four launches through the second layout buffer
    (14, 18, 16, 0, 0, "FWD", 6, 9, 1),     # composite 2-D plan on sub-plans
    (14, 18, 16, 0, 0, "PAIR", 0, 20, 1),   # composite pair: middle buffer + two sub-plans with scratch of their own
    (13, 32, 16, 1, 0, "INV", 0, 11, 1),    # the generic 64-bit passes
    (17, 16, 16, 1, 0, "FWD", 0, 7, 4),     # long unscaled frames: pre-pass + the two wide passes, two scratch parts, two streams (round 5)
    (17, 24, 24, 1, 0, "FWD", 0, 5, 6),     # ... with the 64-bit first pass (24 bytes of scratch per sample)
    (17, 16, 16, 1, 0, "INV", 0, 5, 6),     # ... the inverse: gather pass, block pass, 64-bit post-pass
    (17, 24, 24, 1, 0, "INV", 0, 3, 8),     # ... the inverse on 64-bit words throughout (32 bytes of scratch per sample)
    (18, 18, 18, 0, 0, "FWD", 0, 5, 4),     # long general-width frames: pre-pass + pass A in place + pass B (round 5)
    (17, 18, 18, 0, 1, "INV", 0, 7, 2),     # ... and the inverse
]


@pytest.mark.parametrize("case", WS_CASES)
def test_every_plan_is_reentrant_on_caller_workspaces(case, monkeypatch):
    """intfft_exec_ws: ONE plan, two streams at once, each call with its own workspace -- bit-exact against the oracle and against
    intfft_exec on the plan's own scratch.  Then: an under-sized workspace (sub-batches), a workspace that cannot serve one frame (refused),
    overlap with the data (refused), and the plan with its own scratch released (intfft_exec refuses, intfft_exec_ws still runs)."""
    import torch

    from intfftk_amd import IntFFTCore
    from intfftk_amd import _capi as capi

    log2n, dw, tw, fmt, rnd, direction, l1, batch, mb = case
    monkeypatch.setenv("INTFFT_SCRATCH_MB", str(mb))  # small chunks: the chunk loops (and the two-stream alternation) run on small batches
    n = 1 << log2n
    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW", direction, "NATURAL", "NATURAL", NFFT1=l1)
    assert core.info["scratch_bytes"] > 0, core.info
    need = core.workspace_bytes(batch)
    assert 0 < core.workspace_bytes(1) <= need <= core.workspace_bytes(10 * batch) and need % 256 == 0
    npdt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
    xs = [uniform_frames(batch, n, dw - 1, 700 + 13 * i + log2n) for i in range(2)]
    xd = [torch.from_numpy(x.astype(npdt)).cuda() for x in xs]
    p = C.make_params(log2n, dw, tw, fmt, rnd, True)
    d = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}[direction]
    want = [C.execute_2d(x, p, l1, d) if l1 else C.execute(x, p, d) for x in xs]
    own = [core(x) for x in xd]
    for o, w in zip(own, want):
        assert np.array_equal(o.cpu().numpy().astype(np.int64), w)
    wss = [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(2)]
    ys = [torch.zeros_like(o) for o in own]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    for rep_ in range(3):  # interleaved calls, no host synchronisation in between
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                core.exec_ws(xd[i], wss[i], out=ys[i])
    torch.cuda.synchronize()
    for y, o in zip(ys, own):
        assert torch.equal(y, o)
    # an under-sized workspace: the largest sub-batches it serves; one byte short of a frame: refused
    small = core.workspace_bytes(max(1, batch // 3))
    if small < need:
        y = core.exec_ws(xd[0], torch.empty(small, dtype=torch.uint8, device="cuda"))
        assert torch.equal(y, own[0])
    one = core.workspace_bytes(1)
    with pytest.raises(capi.IntFFTError) as ei:
        core.exec_ws(xd[0], torch.empty(one - 256, dtype=torch.uint8, device="cuda"))
    assert ei.value.status == capi.ERR_INVALID
    with pytest.raises(capi.IntFFTError):
        core.exec_ws(xd[0], None)
    st = capi.lib().intfft_exec_ws(core._plan, xd[0].data_ptr(), ys[0].data_ptr(), batch, xd[0].data_ptr(), need, None)
    assert st == capi.ERR_INVALID  # workspace overlaps the input
    st = capi.lib().intfft_exec_ws(core._plan, xd[0].data_ptr(), ys[0].data_ptr(), batch, wss[0].data_ptr() + 4, need - 4, None)
    assert st == capi.ERR_INVALID  # not 256-byte aligned
    # the plan without its own scratch
    core.release_scratch()
    assert core.info["scratch_bytes"] == 0
    with pytest.raises(capi.IntFFTError) as ei:
        core(xd[0])
    assert ei.value.status == capi.ERR_INVALID
    ys[1].zero_()
    core.exec_ws(xd[1], wss[1], out=ys[1])
    assert torch.equal(ys[1], own[1])
    core.close()


def test_exec_ws_on_single_launch_plans_needs_no_workspace():
    import torch

    from intfftk_amd import IntFFTCore

    core = IntFFTCore(10, 16, 16, 0, 0, "NEW", "FWD", "NATURAL", "NATURAL")
    assert core.workspace_bytes(4096) == 0
    x = torch.from_numpy(uniform_frames(33, 1024, 15, 4).astype(np.int16)).cuda()
    assert torch.equal(core.exec_ws(x), core(x))
    core.release_scratch()  # nothing to free, but the switch is the same for every plan: only exec_ws runs it from here on
    assert torch.equal(core.exec_ws(x, None), core.exec_ws(x))
    core.close()


def test_exec_ws_from_two_host_threads():
    """Two host threads, one plan (two-stream N = 2^20 plan: the side stream comes from the plan's pool), each thread with its own stream and
    workspace, 6 calls each: same bits as the plan's own scratch."""
    import threading

    import torch

    from intfftk_amd import IntFFTCore

    os.environ["INTFFT_SCRATCH_MB"] = "8"
    try:
        core = IntFFTCore(20, 16, 16, 0, 0, "NEW", "FWD", "NATURAL", "NATURAL")
    finally:
        del os.environ["INTFFT_SCRATCH_MB"]
    batch = 7
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    xs = [torch.randint(-(1 << 14), 1 << 14, (batch, 1 << 20, 2), device="cuda", dtype=torch.int16, generator=g) for _ in range(2)]
    want = [core(x) for x in xs]
    need = core.workspace_bytes(batch)
    ys = [torch.zeros_like(w) for w in want]
    wss = [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    core.exec_ws(xs[i], wss[i], out=ys[i])
            st.synchronize()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for y, w in zip(ys, want):
        assert torch.equal(y, w)
    core.close()


@pytest.mark.parametrize("pieces", [None, 1, 3, 8])
def test_exec_sharded_async_equals_blocking(pieces, monkeypatch):
    """intfft_exec_sharded_async: no host synchronisation -- the input is produced on the caller's stream right before the call and the
    result consumed on it right after; shards move in pieces (INTFFT_SHARD_PIECES; default: up to 4 of >= 2 MiB) with the copy back of piece
    k beside the copy in of piece k + 1.  Bit-equal to the blocking call and the oracle; three plans per visible device (on a single-GPU box
    all of them share it: the peer copies become local copies, the schedule is the same), ragged batch, back-to-back calls, another root."""
    import torch

    from intfftk_amd import IntFFTCore, exec_sharded
    from intfftk_amd import _capi as capi

    if pieces is not None:
        monkeypatch.setenv("INTFFT_SHARD_PIECES", str(pieces))
    ndev = torch.cuda.device_count()
    nplans = max(3, ndev)
    cores = [IntFFTCore(12, 16, 16, 0, 0, "NEW", "PAIR", device=i % ndev) for i in range(nplans)]
    arr = (ctypes.c_void_p * nplans)(*[c._plan for c in cores])
    batch = 1031  # 1031 = 343 + 344 + 344 frames of 16 KiB: 5.4 MiB per shard
    assert capi.lib().intfft_shard_prepare(arr, nplans, 0, batch) == 0
    x = uniform_frames(batch, 4096, 15, 21)
    want = C.execute(x, C.make_params(12, 16, 16, 0, 0, True), C.PAIR)
    xh = torch.from_numpy(x.astype(np.int16)).pin_memory()
    y_block = exec_sharded(cores, xh.to("cuda:0"), 0)
    assert np.array_equal(y_block.cpu().numpy().astype(np.int64), want)
    s = torch.cuda.Stream(device=0)
    with torch.cuda.stream(s):
        outs = []
        for rep_ in range(3):  # produced on s, transformed behind it, consumed on s: no synchronisation anywhere
            xd = xh.to("cuda:0", non_blocking=True)
            xd2 = xd + 0  # a kernel on s the call has to wait for
            y = exec_sharded(cores, xd2, 0, asynchronous=True)
            outs.append(y.clone())  # a kernel on s that has to wait for the call
    s.synchronize()
    for y in outs:
        assert torch.equal(y, y_block)
    root = 1
    xd = xh.to("cuda:%d" % (root % ndev))
    torch.cuda.synchronize()
    y1 = exec_sharded(cores, xd, root, asynchronous=True)
    torch.cuda.synchronize()
    assert torch.equal(y1.cpu(), y_block.cpu())
    for c in cores:
        c.close()


GUARD_CASES = [
    # (log2n, dw, tw, fmt, rnd, direction, in_order, out_order, batch): every dedicated kernel family at a batch that does not fill its
    # last chunk / group / tile, so the predicated stores of the partial paths are what runs
    (3, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 67), (5, 16, 16, 0, 1, "PAIR", "NATURAL", "NATURAL", 131),
    (6, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 17), (7, 16, 16, 0, 0, "INV", "BITREV", "HALVES", 9), (9, 16, 16, 0, 0, "PAIR", "NATURAL", "NATURAL", 5),
    (10, 16, 16, 0, 0, "FWD", "HALVES", "BITREV", 7), (10, 12, 16, 0, 1, "FWD", "NATURAL", "NATURAL", 3), (7, 16, 16, 1, 0, "FWD", "NATURAL", "NATURAL", 11),
    (8, 16, 16, 1, 0, "PAIR", "NATURAL", "NATURAL", 7), (9, 24, 18, 1, 0, "FWD", "NATURAL", "NATURAL", 3), (10, 18, 16, 0, 0, "INV", "NATURAL", "NATURAL", 5),
    (11, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 3), (11, 16, 16, 0, 0, "PAIR", "NATURAL", "NATURAL", 5), (12, 16, 16, 0, 1, "INV", "NATURAL", "NATURAL", 3),
    (11, 24, 24, 1, 0, "FWD", "NATURAL", "NATURAL", 3), (12, 16, 16, 1, 0, "INV", "NATURAL", "NATURAL", 3), (7, 32, 16, 1, 0, "FWD", "NATURAL", "NATURAL", 13),
    (11, 32, 16, 1, 0, "INV", "NATURAL", "NATURAL", 3), (13, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 5), (14, 16, 16, 0, 0, "PAIR", "NATURAL", "NATURAL", 3),
    (13, 16, 16, 0, 1, "FWD", "HALVES", "BITREV", 5), (15, 16, 16, 0, 0, "INV", "NATURAL", "NATURAL", 3), (16, 16, 16, 0, 0, "PAIR", "NATURAL", "NATURAL", 3),
    (13, 24, 24, 1, 0, "FWD", "NATURAL", "NATURAL", 3), (14, 24, 24, 1, 0, "INV", "NATURAL", "NATURAL", 5), (13, 16, 16, 1, 0, "FWD", "NATURAL", "NATURAL", 3),
    (17, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 3), (18, 16, 16, 0, 1, "INV", "NATURAL", "NATURAL", 1), (19, 16, 16, 0, 0, "FWD", "NATURAL", "BITREV", 3),
    (20, 16, 16, 0, 0, "INV", "NATURAL", "NATURAL", 1), (10, 18, 16, 0, 0, "PAIR", "NATURAL", "NATURAL", 3), (10, 40, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 3),
    (9, 16, 16, 0, 0, "FWD", "NATURAL", "BITREV_LANES", 3),
    # round 5: the 64-bit first pass (DATA_WIDTH 25 .. 32), the cores' own orders on the two-pass wide / general-width kernels (16- and 32-byte pair accesses,
    # the 16 x 16 exchange), the full-line column tile of N = 2^19
    (13, 32, 16, 1, 0, "FWD", "NATURAL", "NATURAL", 11), (14, 30, 16, 1, 0, "INV", "NATURAL", "NATURAL", 5), (16, 24, 24, 1, 0, "FWD", "HALVES", "BITREV", 3),
    (14, 24, 24, 1, 0, "INV", "BITREV", "HALVES", 5), (13, 32, 16, 1, 0, "FWD", "HALVES", "BITREV", 3), (15, 32, 16, 1, 0, "INV", "BITREV", "HALVES", 3),
    (14, 16, 16, 1, 0, "FWD", "HALVES", "BITREV", 5), (15, 18, 16, 0, 0, "INV", "BITREV", "HALVES", 3), (13, 12, 16, 0, 1, "FWD", "HALVES", "NATURAL", 9),
    (19, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 3),
    # ... the cores' own orders on the 64-bit wave / block kernels (N = 2048: the absent second frame of the last workgroup; N < 1024: absent frames of the last wave)
    (10, 32, 16, 1, 0, "FWD", "HALVES", "BITREV", 7), (10, 40, 16, 0, 0, "INV", "BITREV", "HALVES", 5), (12, 32, 16, 1, 0, "FWD", "HALVES", "BITREV", 3),
    (11, 32, 16, 1, 0, "INV", "BITREV", "HALVES", 3), (11, 40, 16, 0, 0, "FWD", "HALVES", "BITREV", 5), (8, 32, 16, 1, 0, "FWD", "HALVES", "BITREV", 13),
    (9, 32, 16, 1, 0, "INV", "BITREV", "HALVES", 5),
    # ... and the 2-D plans of round 5 (a tenth field: log2 N1): the inverse and the pair at N = 2^21, the three-launch inverse at N = 2^22
    (21, 16, 16, 0, 0, "INV", "NATURAL", "NATURAL", 1, 10), (21, 16, 16, 0, 0, "PAIR", "NATURAL", "NATURAL", 1, 10), (22, 16, 16, 0, 0, "INV", "NATURAL", "HALVES", 1, 10),
    (21, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 1, 10), (22, 16, 16, 0, 0, "FWD", "NATURAL", "NATURAL", 1, 11), (22, 16, 16, 0, 0, "INV", "NATURAL", "NATURAL", 1, 11),
    # ... the long frames outside 16-bit scaled data (round 5): pre- / post-pass pair accesses, the last / first pass with its rows across the blocks
    (17, 16, 16, 1, 0, "FWD", "NATURAL", "NATURAL", 3), (18, 16, 16, 1, 0, "FWD", "HALVES", "BITREV", 1), (17, 24, 24, 1, 0, "FWD", "HALVES", "BITREV", 1),
    (17, 16, 16, 1, 0, "INV", "BITREV", "HALVES", 1), (17, 24, 24, 1, 0, "INV", "NATURAL", "NATURAL", 1), (17, 18, 18, 0, 0, "FWD", "HALVES", "BITREV", 3),
    (18, 18, 18, 0, 1, "INV", "BITREV", "HALVES", 1), (17, 12, 16, 1, 0, "FWD", "NATURAL", "NATURAL", 3),
]


@pytest.mark.parametrize("shift_samples", [0, 1])
@pytest.mark.parametrize("case", GUARD_CASES)
def test_no_writes_outside_the_output_buffer(case, shift_samples):
    """Buffer-overrun check of every kernel family on ragged batches: the output lives between two poisoned guard bands of one allocation (and
    the input between two others, so a read beyond it would at least read poison, not a neighbour's data); after intfft_exec the bands still
    hold the poison and the result equals the oracle's.  shift_samples = 1: both buffers start ONE complex sample behind a 64 KiB boundary (the ABI asks
    for container alignment only: the 16-byte vector accesses of the kernels must not assume more)."""
    import torch

    from intfftk_amd import IntFFTCore

    log2n, dw, tw, fmt, rnd, direction, in_o, out_o, batch = case[:9]
    l1 = case[9] if len(case) > 9 else 0
    if not l1 and C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, True), {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}[direction]) != 0:
        pytest.skip("not elaboratable")
    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW", direction, in_o, out_o, NFFT1=l1)
    n = 1 << log2n
    x = uniform_frames(batch, n, dw, 7700 + log2n + dw)
    xin = torch.from_numpy(np.ascontiguousarray(x.astype({2: np.int16, 4: np.int32, 8: np.int64}[core.in_container])))
    guard = 1 << 16  # bytes on each side
    gi, go = guard + shift_samples * 2 * core.in_container, guard + shift_samples * 2 * core.out_container
    in_bytes, out_bytes = batch * n * 2 * core.in_container, batch * n * 2 * core.out_container
    ibuf = torch.full((gi + in_bytes + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    obuf = torch.full((go + out_bytes + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    ibuf[gi:gi + in_bytes] = xin.cuda().view(torch.uint8).reshape(-1)
    torch.cuda.synchronize()
    core.exec_raw(ibuf.data_ptr() + gi, obuf.data_ptr() + go, batch, 0)
    torch.cuda.synchronize()
    info = core.info
    assert bool((obuf[:go] == 0xA5).all()) and bool((obuf[go + out_bytes:] == 0xA5).all()), ("write outside the output buffer", info)
    assert bool((ibuf[:gi] == 0x5A).all()) and bool((ibuf[gi + in_bytes:] == 0x5A).all()), ("write into the input's guard bands", info)
    got = obuf[go:go + out_bytes].clone().view(core.out_dtype).reshape(core.out_shape(batch)).cpu().numpy().astype(np.int64)
    p = C.make_params(log2n, dw, tw, fmt, rnd, True)
    om = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
    d = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}[direction]
    want = C.execute_2d(x, p, l1, d, om[in_o], om[out_o], form=1) if l1 else C.execute(x, p, d, om[in_o], om[out_o], form=1)
    assert np.array_equal(got, want), info
    core.close()


def test_exec_under_stream_capture():
    """intfft_exec inside a hipGraph capture: every launch stays on the capturing stream (the chunked plans do not fork onto their side
    stream there), and the replayed graph reproduces the eager result -- a multi-chunk two-pass plan and the headline kernel."""
    import torch

    from intfftk_amd import IntFFTCore

    for log2n, batch, mb in ((20, 9, "16"), (10, 4096, None)):
        if mb:
            os.environ["INTFFT_SCRATCH_MB"] = mb
        try:
            core = IntFFTCore(log2n, 16, 16, 0, 0, "NEW", "FWD", "NATURAL", "NATURAL")
        finally:
            os.environ.pop("INTFFT_SCRATCH_MB", None)
        n = 1 << log2n
        g = torch.Generator(device="cuda")
        g.manual_seed(11 + log2n)
        x = torch.randint(-(1 << 14), 1 << 14, (batch, n, 2), device="cuda", dtype=torch.int16, generator=g)
        want = core(x)
        y = torch.zeros_like(want)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(graph, stream=s):
            rc = core.exec_raw(x.data_ptr(), y.data_ptr(), batch, torch.cuda.current_stream().cuda_stream)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        core.close()
