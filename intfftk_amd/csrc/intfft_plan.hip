// intfft_plan.hip -- host side of libintfft.so: elaboration checks, twiddle tables, pass planning
// and the C-ABI of include/intfft.h.
//
// The plan plays the role of RTL elaboration: it accepts exactly the generic combinations for
// which int_fftNk / int_ifftNk elaborate (find_delay != 0, src/vhdl/fft/int_dif2_fly.vhd:87-116;
// regime conditions src/vhdl/math/cmult/int_cmult_dsp48.vhd:182-434) and fixes the per-stage
// widths DATA_WIDTH + ii*FORMAT (src/vhdl/fft/int_fftNk.vhd:187-207).
//
// There is no CPU execution path in this library: without a HIP device every entry point that
// would compute returns INTFFT_ERR_NO_DEVICE.
#include "../../include/intfft.h"
#include "intfft_internal.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

namespace intfft {

namespace {
std::mutex g_geom_mu;
std::map<std::pair<const void *, int>, int> g_occupancy; // (kernel, device) -> resident blocks per CU (0: query failed)
std::map<std::pair<const void *, int>, bool> g_max_lds;
std::map<int, int> g_cus;

int current_device()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}

int cus_locked(int dev)
{
    auto it = g_cus.find(dev);
    if (it != g_cus.end()) return it->second;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_cus[dev] = cus;
    return cus;
}
} // namespace

int device_cus()
{
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_geom_mu);
    return cus_locked(dev);
}

size_t resident_blocks(const void *kernel, int threads, int dflt, int max_per_cu, bool env_override)
{
    static const int env = diag_env("INTFFT_BLOCKS_PER_CU") ? atoi(diag_env("INTFFT_BLOCKS_PER_CU")) : 0;
    const int dev = current_device();
    int per_cu, cus;
    {
        std::lock_guard<std::mutex> lk(g_geom_mu);
        cus = cus_locked(dev);
        const auto key = std::make_pair(kernel, dev);
        auto it = g_occupancy.find(key);
        if (it == g_occupancy.end()) {
            int q = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kernel, threads, 0) != hipSuccess || q < 0) q = 0;
            it = g_occupancy.emplace(key, q).first;
        }
        per_cu = it->second;
    }
    if (per_cu <= 0) per_cu = dflt;
    if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;
    if (env_override && env > 0) per_cu = env;
    return (size_t)cus * (size_t)per_cu;
}

void allow_max_lds(const void *kernel)
{
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_geom_mu);
    bool &done = g_max_lds[std::make_pair(kernel, dev)];
    if (done) return;
    // 160 KiB minus the kernel's static LDS (e.g. the 256 bytes __syncthreads_or uses): asking for more than fits fails
    // with hipErrorInvalidValue and would leave that error pending for the next hipGetLastError()
    hipFuncAttributes fa;
    int room = 160 * 1024;
    if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) room -= (int)fa.sharedSizeBytes;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room) != hipSuccess) (void)hipGetLastError();
    done = true;
}

} // namespace intfft

using namespace intfft;

constexpr int SHARD_EVENTS = 3 + 8; // entry, done, first-stream done, one per piece of a shard

struct ExecCtx {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

struct intfft_plan {
    intfft_params p;
    int device = 0;
    int L = 0;
    int in_bits = 0, out_bits = 0, in_cb = 0, out_cb = 0;
    int word = 4; // bytes of the on-chip word: 4 / 8 / 16 = k_pass<int32 / int64 / __int128>, 2 = packed int16 (k_pass16)
    int l1 = 0;               // 2-D scheme (intfft_plan_create_2d): log2 N1 of the column core; 0 = ordinary 1-D plan
    // 2-D scheme, composite form (the default): layout change / column cores / layout change + multiply / row cores / layout
    // change, every core an ordinary 1-D sub-plan in natural order (so every dedicated kernel applies); see exec_2d()
    struct intfft_plan *sub_col_f = nullptr, *sub_row_f = nullptr, *sub_row_i = nullptr, *sub_col_i = nullptr;
    StageDesc tw_f{}, tw_i{};  // the multiplier between the cores (widths / regime), forward and inverse
    void *buf2d[2] = {nullptr, nullptr};
    int fused2d = 0;                  // 2: N = 2^20 = 1024 x 1024, 16-bit scaled-truncate forward in two launches (k_big2x_c + k_big2x_b); 3: N = 2^21 .. 2^24 as
                                      // 1024 x N2: k_big2x_c, the row sub-plan, one layout change; 4: N = 2^20 inverse in two launches (k_big2x_qb + k_big2x_ci);
                                      // 5: N = 2^20 pair = the forward two launches, then the inverse two; 6 (round 5): N = 2^21 inverse in two launches
                                      // (k_rows2k_qtr + k_big2x_ci<., 11>); 7: N = 2^21 pair = k_big2x_c<11> + k_rows2k_tr, then form 6's two launches;
                                      // 8 (round 5): N = 2^22 = 2048 x 2048 forward in two launches (k_cols2k_c + k_rows2k_tr<., 11>);
                                      // 10 / 11 (round 5): N = 2^22 = 2048 x 2048 inverse in two launches (k_rows2k_qtr<., 11, true> + k_cols2k_ci) / the pair in four;
                                      // 9 (round 5): N = 2^22 .. 2^24 inverse as one layout change, the N2-point row sub-plan, k_big2x_ci<., L2, ROWS>
    uint2 *d_tw16r = nullptr, *d_tw16ri = nullptr; // fused2d == 3 at N2 = 2048, natural order out (round 5): the row core's packed tables for k_rows2k_tr
    uint32_t *d_tw2d_tiles = nullptr; // its inter-core twiddle table, [chunk][rho][16 columns] of (wr | wi << 16)
    size_t buf2d_frames = 0;
    int2 *d_tw2d = nullptr;   // 2-D scheme: the inter-pass table W_N^m, N entries
    std::vector<int2> h_tw2d;
    int2 *d_tw = nullptr;
    uint2 *d_tw16f = nullptr, *d_tw16i = nullptr; // packed dot-product operand forms (k_pass16)
    std::vector<int2> h_tw;
    std::vector<PassArgs> passes;
    void *d_scratch = nullptr;
    // Multi-pass plans of the dedicated kernels process a batch in scratch-sized chunks; consecutive chunks alternate between the
    // caller's stream and a plan-owned side stream, each with its own scratch half, so that a pass of one chunk runs beside another pass
    // of the next (fork / join with events: the call stays asynchronous on the caller's stream).  Measured: C4 289 against 269
    // Gsample/s, C3 142 against 132 (tools/two_stream_probe.py: two 128 MiB halves beat one 256 MiB scratch and three streams);
    // only the plan families that gain have a second half (create_plan).
    void *d_scratch2 = nullptr;
    // The side stream and its fork / join events are an "execution context" taken from a per-plan pool for the duration of one call
    // (host side) and put back when the call returns: concurrent calls on one plan (intfft_exec_ws on distinct workspaces) each get their
    // own, so nothing in the plan is written by an exec.  Re-recording a pooled event later is harmless: a hipStreamWaitEvent already
    // enqueued keeps waiting for the record that preceded it.
    bool wants_side = false;     // this plan's chunk loop alternates between the caller's stream and a side stream
    std::mutex ctx_mu;
    std::vector<ExecCtx> ctx_pool;
    size_t scratch_frames = 0, scratch_bytes = 0;
    size_t scratch_frame_bytes = 0; // bytes of one frame of inter-pass words
    size_t scratch_gran = 1;        // the scratch is used in units of this many frames (the virtual 2^16-point frames of the wide classes)
    bool dual_scratch = false;      // two scratch halves (one per stream)
    bool is2d = false, is_pair = false;
    int n2d_bufs = 0;               // layout buffers a 2-D plan uses (1: the two-launch fused forms)
    bool owns_scratch = true;       // false after intfft_plan_release_scratch: only intfft_exec_ws runs the plan then
    bool fast1024 = false;
    bool fast4096 = false;
    bool fast16k = false;  // N = 8192 / 16384, 16-bit scaled-truncate FWD / INV in ONE pass (intfft_fast16k.hip)
    bool fast1024x = false;
    bool fast1024u = false;
    bool fast1024ux = false;
    bool fastw32 = false;
    // PAIR plans on 64-bit words whose two halves both have dedicated kernels: int_fftNk sub-plan -> middle buffer -> int_ifftNk sub-plan
    // (int_fft_ifft_pair.vhd:209-280: the cores are chained; natural order in the middle is the same chain with both reorders applied)
    intfft_plan *pair_f = nullptr, *pair_i = nullptr;
    void *pair_buf = nullptr;
    size_t pair_frames = 0;
    int lanes_mode = 0;             // BITREV_LANES composite: 1 = pair_f (BITREV out) -> pair_buf -> bit permutation, 2 = bit permutation -> pair_buf -> pair_f (BITREV in);
                                    // 3 = USE_FLY = 0: width conversion (-> pair_buf -> bit permutation bypass_perm unless it is the identity), no sub-plan
    int bypass_perm[24] = {0};      // lanes_mode 3: m_in bit bypass_perm[b] = m_out bit b
    bool fastw64 = false;  // N = 64 .. 1024 forward / inverse, results of 33 .. 64 bits: the 64-bit wave kernel (intfft_fastw64.hip)
    StageDesc st64[12] = {};
    bool fastw64b = false; // N = 2048 / 4096 forward / inverse, results of 33 .. 64 bits beyond k_fft4096_w32's 64-bit last round
    bool fast4096w = false;
    bool w32inv = false;
    bool bigw = false;
    bool fastsmall = false;
    W32Args w32args{};
    UxArgs uxargs{};
    bool big20 = false;
    bool big_pair256 = false;  // N = 2^13 .. 2^16 pair: k_big20_p1<., ., 8>, k_mid_pair, k_big20_q1<., ., 8> (256 x 256 split)
    bool big_two_pass = false; // N = 2^13 .. 2^16 FWD / INV: k_big20_p1<., ., 8> + k_mid_p2 | k_mid_c, k_mid_q1 | k_mid_c + k_big20_q1<., ., 8>
    bool wide16 = false;
    bool widelong = false; // wide16 at N = 2^17 .. 2^20 (three launches, 16-byte scratch samples)
    WideArgs wargs{};
    Fast1024Args fargs{};
    // host-streaming state (intfft_exec_host), created on first use
    hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
    void *slot_in[2] = {nullptr, nullptr}, *slot_out[2] = {nullptr, nullptr};
    // intfft_exec_sharded: this plan's staging buffers (grow only) and stream
    void *shard_in = nullptr, *shard_out = nullptr;
    size_t shard_in_bytes = 0, shard_out_bytes = 0;
    hipStream_t s_shard = nullptr;  // scatter side + transforms of this plan's shard
    hipStream_t s_shard2 = nullptr; // gather side (peer copies) / the root's own transform
    hipStream_t s_call = nullptr;   // the blocking intfft_exec_sharded's "caller stream" (root plan)
    hipEvent_t shard_ev[SHARD_EVENTS] = {};
    bool shard_used = false;        // shard_ev[1] has been recorded by an earlier call
    int shard_peer = -1; // root device this plan's device has peer access to (-1: not set up yet)
    // RCCL transport of intfft_exec_sharded (intfft_shard_set_transport): this plan's communicator of the plan set (rank = its index in the
    // set), the set's size; plans[0] of the set owns all of them (rccl_owned)
    void *rccl_comm = nullptr;
    int rccl_rank = -1, rccl_nranks = 0;
    uint64_t rccl_set = 0;               // which ncclCommInitAll this communicator came from (0: none)
    intfft_plan *rccl_owner = nullptr;   // the plan that owns the set's communicators (plans[0] of the set)
    std::vector<void *> rccl_owned;
    std::vector<intfft_plan *> rccl_members; // owner only: every plan that holds one of rccl_owned (back-pointers for release)
    size_t slot_frames = 0;
    char kernel_name[64] = {0};
};

static void free_stream_state(intfft_plan *pl);

static void ctx_destroy(ExecCtx &c)
{
    if (c.side) (void)hipStreamDestroy(c.side);
    if (c.fork) (void)hipEventDestroy(c.fork);
    if (c.join) (void)hipEventDestroy(c.join);
    c = ExecCtx{};
}

// One execution context for the duration of a call (the caller holds the plan's device current).
static bool ctx_acquire(intfft_plan *pl, ExecCtx &c)
{
    {
        std::lock_guard<std::mutex> lk(pl->ctx_mu);
        if (!pl->ctx_pool.empty()) {
            c = pl->ctx_pool.back();
            pl->ctx_pool.pop_back();
            return true;
        }
    }
    c = ExecCtx{};
    hipError_t e = hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.join, hipEventDisableTiming);
    if (e != hipSuccess) {
        ctx_destroy(c);
        (void)hipGetLastError();
        return false;
    }
    return true;
}

static void ctx_release(intfft_plan *pl, const ExecCtx &c)
{
    std::lock_guard<std::mutex> lk(pl->ctx_mu);
    pl->ctx_pool.push_back(c);
}

// Fork / join of a chunk loop that alternates between the caller's stream and a pooled side stream; joins (and returns the context) on
// every exit path.
struct SideStream {
    intfft_plan *pl;
    hipStream_t user;
    ExecCtx c;
    bool on = false;
    SideStream(intfft_plan *p, hipStream_t s) : pl(p), user(s) {}
    hipError_t fork() // under stream capture the call stays on the caller's stream (no cross-stream edges in somebody else's graph)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(user, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return hipSuccess;
        if (!ctx_acquire(pl, c)) return hipSuccess; // no context: one stream, still correct
        hipError_t e = hipEventRecord(c.fork, user);
        if (e == hipSuccess) e = hipStreamWaitEvent(c.side, c.fork, 0);
        if (e != hipSuccess) {
            ctx_release(pl, c);
            return e;
        }
        on = true;
        return hipSuccess;
    }
    ~SideStream()
    {
        if (!on) return;
        if (hipEventRecord(c.join, c.side) == hipSuccess) (void)hipStreamWaitEvent(user, c.join, 0);
        ctx_release(pl, c);
    }
};

static size_t ws_align(size_t b) { return (b + 255) & ~(size_t)255; }

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int container_bytes(int bits) { return bits <= 16 ? 2 : bits <= 32 ? 4 : bits <= 64 ? 8 : 16; }

// (pre-shift a, post-shift b) of the multiplier regime for data width w / twiddle width t,
// or false where the RTL has no generate branch (int_cmult_dsp48.vhd:182-434).
// clip: trpl18 with MAW beyond the A port (AWD = 61 / 59 bits): the port is SXT(M_AA, AWD), which CUTS a longer operand to
// its low AWD bits (int_cmult_trpl18_dsp48.vhd:161-162; the block is written for "data width from 42/44 to 59/61",
// int_cmult_dsp48.vhd:266).  And its product slice P(MAW+MBW-2 downto MBW-1) (:151-152) must lie inside the PWD = 79 / 77
// bits of P, or the slice does not elaborate: MAW + MBW <= PWD + 1.
bool cmult_shifts(int w, int t, int xser, int &a, int &b, int *clip = nullptr)
{
    const int l18 = xser ? 28 : 26, h18 = xser ? 45 : 43, t18 = xser ? 79 : 77, td = xser ? 28 : 26;
    if (clip) *clip = 0;
    if (t < 19) {
        if (w < l18) { a = 0; b = t - 1; return true; }                         // sngl   :184-225
        if (w < h18) { a = xser ? t - 4 : t - 6; b = xser ? 3 : 5; return a >= 0; } // dbl18  :228-264
        if (w < t18) {                                                          // trpl18 :267-303
            const int pwd = xser ? 79 : 77, awd = xser ? 61 : 59;
            if (w + t > pwd + 1) return false;
            a = t - 1;
            b = 0;
            if (clip && w > awd) *clip = awd;
            return true;
        }
        return false;
    }
    if (t < td) {
        if (w < 19) { a = 0; b = t - 2; return true; }                          // sngl25 :309-354
        if (w < 36) { a = t - 14; b = 12; return true; }                        // dbl35  :357-393
        if (w < 53) { a = t - 2; b = 0; return true; }                          // trpl52 :396-433
    }
    return false;
}

// stage list of one core: FFT stage ii has STAGE = NFFT-ii-1 (int_fftNk.vhd:192), IFFT stage ii has
// STAGE = ii (int_ifftNk.vhd:189); DTW = DATA_WIDTH + ii*FORMAT; DIF multiplies at DTW+1-SCALE
// (int_dif2_fly.vhd:351), DIT at DTW (int_dit2_fly.vhd:307).
int core_stages(const intfft_params &p, int dw_in, bool inverse, std::vector<StageDesc> &out)
{
    const int L = p.log2n;
    for (int ii = 0; ii < L; ++ii) {
        StageDesc st{};
        st.kind = inverse ? KIND_DIT : KIND_DIF;
        st.s = st.ts = inverse ? ii : L - ii - 1;
        st.tshift = 0;
        st.lb = 0;
        st.dtw = dw_in + ii * p.format;
        st.wo = st.dtw + p.format; // DTW - SCALE + 1
        st.mw = inverse ? st.dtw : st.wo;
        st.rnd = p.format ? RND_UNSCALED : (p.rndmode ? RND_ROUND : RND_TRUNC);
        st.sh_a = st.sh_b = 0;
        int clip = 0;
        if (st.s >= 2 && !cmult_shifts(st.mw, p.twdl_width, p.xser, st.sh_a, st.sh_b, &clip))
            return INTFFT_ERR_UNSUPPORTED;
        st.narrow = clip ? clip : (st.mw + p.twdl_width <= 64 && !diag_env("INTFFT_NO_NARROW_MUL")) ? 1 : 0;
        st.tw_off = (1u << st.s) - 1u; // tables of stages 0..s-1 precede: sum 2^i = 2^s - 1
        out.push_back(st);
    }
    return INTFFT_OK;
}

// Stage list of the N > 512K "2-D scheme" (this project's extension of int_fftNk.vhd:11-13):
// N = N1 * N2, forward = N1-point int_fftNk on the top l1 index bits (twiddle index = position div N2), the
// inter-pass multiply by W_N^(k1*n2), then N2-point int_fftNk on the low l2 bits; inverse = the mirror.
int core_stages_2d(const intfft_params &p, int l1, int dw_in, bool inverse, std::vector<StageDesc> &out)
{
    const int L = p.log2n, l2 = L - l1;
    auto butterfly = [&](int ii, int ts, int tshift) -> int {
        StageDesc st{};
        st.kind = inverse ? KIND_DIT : KIND_DIF;
        st.ts = ts;
        st.tshift = tshift;
        st.s = ts + tshift;
        st.dtw = dw_in + ii * p.format;
        st.wo = st.dtw + p.format;
        st.mw = inverse ? st.dtw : st.wo;
        st.rnd = p.format ? RND_UNSCALED : (p.rndmode ? RND_ROUND : RND_TRUNC);
        int clip = 0;
        if (ts >= 2 && !cmult_shifts(st.mw, p.twdl_width, p.xser, st.sh_a, st.sh_b, &clip)) return INTFFT_ERR_UNSUPPORTED;
        st.narrow = clip ? clip : (st.mw + p.twdl_width <= 64) ? 1 : 0;
        st.tw_off = (1u << ts) - 1u;
        out.push_back(st);
        return INTFFT_OK;
    };
    auto twiddle = [&](int w) -> int { // the multiplier instance between the cores, at the width of the data there
        StageDesc st{};
        st.kind = inverse ? KIND_TWMULC : KIND_TWMUL;
        st.s = l2; // lives in the pass that owns index bit l2 (the column core's lowest bit)
        st.ts = 2;
        st.tshift = l2;
        st.dtw = st.wo = st.mw = w;
        st.rnd = RND_UNSCALED;
        int clip = 0;
        if (!cmult_shifts(w, p.twdl_width, p.xser, st.sh_a, st.sh_b, &clip)) return INTFFT_ERR_UNSUPPORTED;
        st.narrow = clip ? clip : (w + p.twdl_width <= 64) ? 1 : 0;
        out.push_back(st);
        return INTFFT_OK;
    };
    int rc;
    if (!inverse) {
        for (int ii = 0; ii < l1; ++ii)
            if ((rc = butterfly(ii, l1 - ii - 1, l2)) != INTFFT_OK) return rc;
        if ((rc = twiddle(dw_in + l1 * p.format)) != INTFFT_OK) return rc;
        for (int ii = 0; ii < l2; ++ii)
            if ((rc = butterfly(l1 + ii, l2 - ii - 1, 0)) != INTFFT_OK) return rc;
    } else {
        for (int ii = 0; ii < l2; ++ii)
            if ((rc = butterfly(ii, ii, 0)) != INTFFT_OK) return rc;
        if ((rc = twiddle(dw_in + l2 * p.format)) != INTFFT_OK) return rc;
        for (int ii = 0; ii < l1; ++ii)
            if ((rc = butterfly(l2 + ii, ii, l2)) != INTFFT_OK) return rc;
    }
    return INTFFT_OK;
}

int validate(const intfft_params &p, int l1 = 0)
{
    if (l1 == 0 && (p.log2n < 3 || p.log2n > 20)) return INTFFT_ERR_INVALID;
    if (l1 != 0 && (l1 < 3 || l1 > 19 || p.log2n - l1 < 3 || p.log2n - l1 > 19 || p.log2n > 24 || !p.use_fly)) return INTFFT_ERR_INVALID;
    if (p.direction < 0 || p.direction > 2) return INTFFT_ERR_INVALID;
    if (p.in_order < 0 || p.in_order > 3 || p.out_order < 0 || p.out_order > 3) return INTFFT_ERR_INVALID;
    if ((p.format | 1) != 1 || (p.rndmode | 1) != 1 || (p.xser | 1) != 1 || (p.use_fly | 1) != 1)
        return INTFFT_ERR_INVALID;
    if (p.data_width < 2 || p.data_width > 64) return INTFFT_ERR_INVALID;
    if (p.twdl_width < 4 || p.twdl_width > 32) return INTFFT_ERR_INVALID;
    // FORMAT=1 with RNDMODE=1 drives wz_re twice: not elaboratable (int_dif2_fly.vhd:339-346)
    if (p.format == 1 && p.rndmode == 1) return INTFFT_ERR_UNSUPPORTED;
    const int growth = p.format ? p.log2n : 0;
    const int out_bits = p.data_width + (p.direction == INTFFT_PAIR ? 2 * growth : growth);
    // results beyond 64 bits (the trpl18 / trpl52 tails with bit growth, int_cmult_dsp48.vhd:267-303, 396-433): 128-bit words
    // and containers in the 1-D plans; the stage check below bounds the widths by what the multiplier elaborates (< 78 bits)
    if (out_bits > (l1 ? 64 : 96)) return INTFFT_ERR_UNSUPPORTED;
    // elaboration check of every stage (find_delay != 0, int_dif2_fly.vhd:87-116) and of the twiddle width
    // (find_twd_25, int_cmult_dsp48.vhd:161-173): intfft_io_widths and intfft_plan_create agree on what elaborates
    if (p.twdl_width >= (p.xser ? 28 : 26)) return INTFFT_ERR_UNSUPPORTED;
    std::vector<StageDesc> tmp;
    int rc = INTFFT_OK;
    auto stages = [&](int dw, bool inverse) { return l1 ? core_stages_2d(p, l1, dw, inverse, tmp) : core_stages(p, dw, inverse, tmp); };
    if (p.direction == INTFFT_FWD || p.direction == INTFFT_PAIR)
        if ((rc = stages(p.data_width, false)) != INTFFT_OK) return rc;
    if (p.direction == INTFFT_INV)
        if ((rc = stages(p.data_width, true)) != INTFFT_OK) return rc;
    if (p.direction == INTFFT_PAIR)
        if ((rc = stages(p.data_width + p.format * p.log2n, true)) != INTFFT_OK) return rc;
    return INTFFT_OK;
}

// How the user-side index map at one end of the transform relates to the core index j:
// true when consecutive memory needs passenger (top) bits in a contiguous-low-bits tile.
bool scatters(int order, bool rev)
{
    // memory = order_to_mem(order, rev ? bitrev(j) : j)
    if (rev) return order == INTFFT_ORDER_NATURAL || order == INTFFT_ORDER_HALVES;
    return order == INTFFT_ORDER_BITREV || order == INTFFT_ORDER_BITREV_LANES;
}

struct Shape {
    int len0, pos0, len1, pos1;
    std::vector<StageDesc> st;
};

int local_bit(const Shape &sh, int s)
{
    if (s >= sh.pos0 && s < sh.pos0 + sh.len0) return s - sh.pos0;
    return sh.len0 + (s - sh.pos1);
}

// Split one core's stages into LDS passes.  `contig_first`: DIT (stages ascend from bit 0);
// otherwise DIF (stages descend from bit L-1).  `passengers`: the contiguous pass also owns that
// many top index bits so that the user-side sweep stays coalesced.
void split_core(int L, int umax, int cmin, const std::vector<StageDesc> &st, bool contig_first,
                int passengers, std::vector<Shape> &out)
{
    std::vector<Shape> shapes;
    const int rf = std::min(L, umax - passengers); // bits of the contiguous pass
    const int strided = L - rf;
    const int per = umax - cmin;
    const int ns = strided > 0 ? (strided + per - 1) / per : 0;
    // contiguous pass: index bits [0, rf) (+ passengers at the top)
    Shape c{};
    c.len0 = rf;
    c.pos0 = 0;
    c.len1 = std::min(passengers, L - rf);
    c.pos1 = L - c.len1;
    // strided passes, from the top of the index downwards
    int hi = L - 1;
    for (int i = 0; i < ns; ++i) {
        const int m = (strided - (L - 1 - hi) + (ns - i) - 1) / (ns - i); // spread evenly
        Shape s{};
        s.len1 = m;
        s.pos1 = hi - m + 1;
        s.len0 = std::min(umax - m, s.pos1);
        s.pos0 = 0;
        shapes.push_back(s);
        hi -= m;
    }
    // attach stages
    auto owns = [](const Shape &sh, int s) {
        return (s >= sh.pos1 && s < sh.pos1 + sh.len1);
    };
    for (const StageDesc &d : st) {
        if (d.s < rf) c.st.push_back(d);
        else
            for (Shape &sh : shapes)
                if (owns(sh, d.s)) sh.st.push_back(d);
    }
    if (contig_first) {
        out.push_back(c);
        for (auto it = shapes.rbegin(); it != shapes.rend(); ++it) out.push_back(*it);
    } else {
        for (Shape &sh : shapes) out.push_back(sh);
        out.push_back(c);
    }
}

bool same_tile(const Shape &a, const Shape &b)
{
    return a.len0 == b.len0 && a.pos0 == b.pos0 && a.len1 == b.len1 && a.pos1 == b.pos1;
}

int build_passes(intfft_plan &pl)
{
    const intfft_params &p = pl.p;
    const int L = pl.L;
    int umax = pl.word == 2 ? 14 : pl.word == 4 ? 13 : pl.word == 8 ? 12 : 11; // 64 KiB tiles
    // packed multi-pass plans: 16 KiB tiles (many workgroups per CU) beat 64 KiB ones (measured on C4)
    if (pl.word == 2 && L > umax) umax = 12;
    if (const char *e = diag_env("INTFFT_TILE_LOG2")) umax = atoi(e) >= 8 && atoi(e) <= umax ? atoi(e) : umax; // diagnostics
    const int cmin = pl.word == 2 ? 6 : pl.word == 4 ? 5 : pl.word == 8 ? 4 : 3; // >= 256 B contiguous per strided row

    std::vector<StageDesc> fwd, inv;
    int rc = INTFFT_OK;
    auto stages = [&](int dw, bool inverse, std::vector<StageDesc> &o) {
        return pl.l1 ? core_stages_2d(p, pl.l1, dw, inverse, o) : core_stages(p, dw, inverse, o);
    };
    if (p.direction == INTFFT_FWD || p.direction == INTFFT_PAIR)
        if ((rc = stages(p.data_width, false, fwd)) != INTFFT_OK) return rc;
    if (p.direction == INTFFT_INV)
        if ((rc = stages(p.data_width, true, inv)) != INTFFT_OK) return rc;
    if (p.direction == INTFFT_PAIR)
        if ((rc = stages(p.data_width + p.format * L, true, inv)) != INTFFT_OK) return rc;

    // user-side maps: FWD out and INV in are on the frequency side (logical = bitrev(core index))
    const bool in_rev = p.direction == INTFFT_INV;
    const bool out_rev = p.direction == INTFFT_FWD;

    std::vector<Shape> shapes;
    if (L <= umax) {
        Shape s{};
        s.len0 = L;
        s.st = fwd;
        s.st.insert(s.st.end(), inv.begin(), inv.end());
        shapes.push_back(s);
    } else {
        if (!fwd.empty()) {
            const int pass = (p.direction == INTFFT_FWD && scatters(p.out_order, out_rev)) ? 4 : 0;
            split_core(L, umax, cmin, fwd, false, pass, shapes);
        }
        if (!inv.empty()) {
            const int pass = (p.direction == INTFFT_INV && scatters(p.in_order, in_rev)) ? 4 : 0;
            std::vector<Shape> is;
            split_core(L, umax, cmin, inv, true, pass, is);
            size_t first = 0;
            if (!shapes.empty() && same_tile(shapes.back(), is[0])) { // fuse DIF tail with DIT head
                shapes.back().st.insert(shapes.back().st.end(), is[0].st.begin(), is[0].st.end());
                first = 1;
            }
            for (size_t i = first; i < is.size(); ++i) shapes.push_back(is[i]);
        }
    }

    for (size_t i = 0; i < shapes.size(); ++i) {
        Shape &sh = shapes[i];
        if ((int)sh.st.size() > MAX_STAGES_PER_PASS) return INTFFT_ERR_UNSUPPORTED;
        PassArgs a{};
        a.L = L;
        a.len0 = sh.len0;
        a.pos0 = sh.pos0;
        a.len1 = sh.len1;
        a.pos1 = sh.len1 ? sh.pos1 : L;
        a.U = sh.len0 + sh.len1;
        // frames per block: enough points that every thread owns work in every round
        int target = pl.word == 2 ? 13 : pl.word == 4 ? 12 : 11; // log2 points per block (int32 words: 11 / 12 / 13 measured 36 / 45 / 47 Gsample/s)
        if (const char *e = diag_env("INTFFT_PASS_TARGET")) target = atoi(e) >= 8 && atoi(e) <= 13 ? atoi(e) : target; // diagnostics
        a.fpb = (a.U == L && L < target) ? (1 << (target - L)) : 1;
        a.in_mode = i == 0 ? IO_USER : IO_SCRATCH;
        a.out_mode = i + 1 == shapes.size() ? IO_USER : IO_SCRATCH;
        a.in_cb = pl.in_cb;
        a.out_cb = pl.out_cb;
        a.in_order = p.in_order;
        a.out_order = p.out_order;
        a.in_rev = in_rev;
        a.out_rev = out_rev;
        a.in_bits = p.data_width;
        a.in_zext = (!p.use_fly && p.format) ? 1 : 0;
        a.ld_swap = (a.in_mode == IO_USER && sh.len1 && scatters(p.in_order, in_rev)) ? 1 : 0;
        a.st_swap = (a.out_mode == IO_USER && sh.len1 && scatters(p.out_order, out_rev)) ? 1 : 0;
        a.ld_memorder = (a.in_mode == IO_USER && a.U == L) ? 1 : 0;
        a.st_memorder = (a.out_mode == IO_USER && a.U == L) ? 1 : 0;
        a.nstages = 0;
        if (p.use_fly) {
            for (StageDesc d : sh.st) {
                d.lb = local_bit(sh, d.s);
                a.st[a.nstages++] = d;
            }
        }
        a.word = a.scr_in_word = pl.word;
        pl.passes.push_back(a);
    }
    // Mixed words: bit growth is monotonic, so the leading passes of a 64-bit plan may still fit 32-bit words
    // (C3: the first 8 of 16 stages have widths <= 32).  Those passes run k_pass<int32> and write 8-byte
    // scratch samples; the first 64-bit pass widens them on load.  Scratch is reused in place, so this needs the
    // first 64-bit pass to be the last pass (it reads 8-byte samples and writes the user array).
    if (pl.word == 8 && pl.passes.size() > 1 && !diag_env("INTFFT_NO_MIXED_WORDS")) {
        size_t k = 0;
        for (; k < pl.passes.size(); ++k) {
            const PassArgs &a = pl.passes[k];
            bool fits = a.nstages > 0 && a.out_mode == IO_SCRATCH && !a.in_zext && (k > 0 || a.in_bits <= 32);
            for (int i = 0; i < a.nstages && fits; ++i) fits = a.st[i].dtw <= 32 && a.st[i].wo <= 32;
            if (!fits) break;
        }
        if (k + 1 == pl.passes.size()) {
            for (size_t i = 0; i < k; ++i) pl.passes[i].word = 4, pl.passes[i].scr_in_word = 4;
            pl.passes[k].scr_in_word = 4;
        }
    }
    return INTFFT_OK;
}

// the inter-pass table of the 2-D scheme (tw2d_eval, intfft_device.hpp) as a device array: only the flat form on the generic
// kernels (INTFFT_2D_GENERIC=1, the A/B reference of the composite form) reads a table; the shipped plans evaluate on the fly
int build_twiddles_2d(intfft_plan &pl)
{
    const size_t n = (size_t)1 << pl.L;
    pl.h_tw2d.resize(n);
    for (size_t m = 0; m < n; ++m) tw2d_eval(pl.L, pl.p.twdl_width, (unsigned)m, pl.h_tw2d[m].x, pl.h_tw2d[m].y);
    hipError_t e = hipMalloc((void **)&pl.d_tw2d, n * sizeof(int2));
    if (e == hipSuccess) e = hipMemcpy(pl.d_tw2d, pl.h_tw2d.data(), n * sizeof(int2), hipMemcpyHostToDevice);
    pl.h_tw2d.clear();
    pl.h_tw2d.shrink_to_fit();
    return (int)e;
}

int build_twiddles(intfft_plan &pl, hipStream_t stream)
{
    const int L = pl.l1 ? std::max(pl.l1, pl.L - pl.l1) : pl.L, t = pl.p.twdl_width;
    const size_t total = ((size_t)1 << L) - 1;
    // quarter-wave ROM seeds, DEPTH = 9: rom_twiddle_int.vhd:135-159
    std::vector<int2> rom(512);
    const double mg = (t < 18) ? std::ldexp(1.0, t - 1) - 1.0 : std::ldexp(1.0, t - 2) - 1.0;
    for (int ii = 0; ii < 512; ++ii) {
        const double phi = ((double)ii * M_PI) / 1024.0;
        rom[ii].x = (int)std::llround(mg * std::cos(phi));
        rom[ii].y = (int)std::llround(mg * std::sin(-phi));
    }
    int2 *d_rom = nullptr;
    hipError_t e;
    if ((e = hipMalloc((void **)&d_rom, rom.size() * sizeof(int2))) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&pl.d_tw, (total + 1) * sizeof(int2))) != hipSuccess) {
        (void)hipFree(d_rom);
        return (int)e;
    }
    e = hipMemcpyAsync(d_rom, rom.data(), rom.size() * sizeof(int2), hipMemcpyHostToDevice, stream);
    for (int s = 0; s < L && e == hipSuccess; ++s)
        e = launch_twiddle_stage(d_rom, s, t, pl.p.xser, pl.d_tw + ((1u << s) - 1u), stream);
    pl.h_tw.resize(total);
    if (e == hipSuccess)
        e = hipMemcpyAsync(pl.h_tw.data(), pl.d_tw, total * sizeof(int2), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_rom);
    return (int)e;
}

} // namespace

// ---- RCCL, loaded at run time (no link-time dependency; the peer-copy transport needs none of it) --------------------------------------
namespace {
struct Rccl {
    typedef int (*comm_init_all_t)(void **, int, const int *);
    typedef int (*comm_destroy_t)(void *);
    typedef int (*group_t)(void);
    typedef int (*sendrecv_t)(const void *, size_t, int, int, void *, hipStream_t);
    typedef int (*recv_t)(void *, size_t, int, int, void *, hipStream_t);
    void *handle = nullptr;
    comm_init_all_t comm_init_all = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    group_t group_start = nullptr, group_end = nullptr;
    sendrecv_t send = nullptr;
    recv_t recv = nullptr;
    bool ok = false;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if ((r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if (!r.handle) return;
        r.comm_init_all = reinterpret_cast<Rccl::comm_init_all_t>(dlsym(r.handle, "ncclCommInitAll"));
        r.comm_destroy = reinterpret_cast<Rccl::comm_destroy_t>(dlsym(r.handle, "ncclCommDestroy"));
        r.group_start = reinterpret_cast<Rccl::group_t>(dlsym(r.handle, "ncclGroupStart"));
        r.group_end = reinterpret_cast<Rccl::group_t>(dlsym(r.handle, "ncclGroupEnd"));
        r.send = reinterpret_cast<Rccl::sendrecv_t>(dlsym(r.handle, "ncclSend"));
        r.recv = reinterpret_cast<Rccl::recv_t>(dlsym(r.handle, "ncclRecv"));
        r.ok = r.comm_init_all && r.comm_destroy && r.group_start && r.group_end && r.send && r.recv;
    });
    return r;
}
constexpr int NCCL_INT8 = 0; // ncclInt8 / ncclChar (rccl.h)
} // namespace

// The owner of a set releases it: every member (whichever sets it is passed in later) forgets its communicator first.
static void rccl_release(intfft_plan *owner)
{
    Rccl &r = rccl();
    for (intfft_plan *m : owner->rccl_members)
        if (m && m->rccl_owner == owner) m->rccl_comm = nullptr, m->rccl_rank = -1, m->rccl_nranks = 0, m->rccl_set = 0, m->rccl_owner = nullptr;
    owner->rccl_members.clear();
    for (void *c : owner->rccl_owned)
        if (c && r.ok) (void)r.comm_destroy(c);
    owner->rccl_owned.clear();
}

// A member leaves its set (destroyed, or moved to another set): the owner keeps the communicator object until it releases the set,
// but no longer points at this plan.
static void rccl_leave(intfft_plan *pl)
{
    if (pl->rccl_owner && pl->rccl_owner != pl)
        for (intfft_plan *&m : pl->rccl_owner->rccl_members)
            if (m == pl) m = nullptr;
    if (pl->rccl_owner != pl) pl->rccl_comm = nullptr, pl->rccl_rank = -1, pl->rccl_nranks = 0, pl->rccl_set = 0, pl->rccl_owner = nullptr;
}

static void rccl_report(const char *what, int status)
{
    if (diag_env("INTFFT_VERBOSE")) std::fprintf(stderr, "intfft: %s failed with ncclResult_t %d\n", what, status);
}

extern "C" {

int intfft_io_widths(const intfft_params *p, int *in_bits, int *out_bits, int *in_cb, int *out_cb)
{
    if (!p) return INTFFT_ERR_NULL;
    const int rc = validate(*p);
    if (rc != INTFFT_OK) return rc;
    const int growth = p->format ? p->log2n : 0;
    const int ob = p->data_width + (p->direction == INTFFT_PAIR ? 2 * growth : growth);
    if (in_bits) *in_bits = p->data_width;
    if (out_bits) *out_bits = ob;
    if (in_cb) *in_cb = container_bytes(p->data_width);
    if (out_cb) *out_cb = container_bytes(ob);
    return INTFFT_OK;
}

static int create_plan(intfft_plan **out, const intfft_params *p, int l1, int hip_device);

int intfft_plan_create(intfft_plan **out, const intfft_params *p, int hip_device)
{
    return create_plan(out, p, 0, hip_device);
}

int intfft_plan_create_2d(intfft_plan **out, const intfft_params *p, int log2_n1, int hip_device)
{
    if (log2_n1 == 0) return INTFFT_ERR_INVALID;
    return create_plan(out, p, log2_n1, hip_device);
}

static int create_plan(intfft_plan **out, const intfft_params *p, int l1, int hip_device)
{
    if (!out || !p) return INTFFT_ERR_NULL;
    *out = nullptr;
    int rc = validate(*p, l1);
    if (rc != INTFFT_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hip_device < 0 || hip_device >= ndev)
        return INTFFT_ERR_NO_DEVICE;
    DeviceGuard guard(hip_device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;

    intfft_plan *pl = new (std::nothrow) intfft_plan();
    if (!pl) return INTFFT_ERR_ALLOC;
    pl->p = *p;
    pl->device = hip_device;
    pl->L = p->log2n;
    pl->l1 = l1;
    {
        const int growth = p->format ? p->log2n : 0;
        pl->in_bits = p->data_width;
        pl->out_bits = p->data_width + (p->direction == INTFFT_PAIR ? 2 * growth : growth);
        pl->in_cb = container_bytes(pl->in_bits);
        pl->out_cb = container_bytes(pl->out_bits);
    }
    pl->word = pl->out_bits <= 32 ? 4 : pl->out_bits <= 64 ? 8 : 16;
    if (!l1 && pass16_supported(p->data_width, p->twdl_width, p->format, p->use_fly)) pl->word = 2;

    if ((rc = build_twiddles(*pl, nullptr)) != INTFFT_OK ||
        (l1 && diag_env("INTFFT_2D_GENERIC") && (rc = build_twiddles_2d(*pl)) != INTFFT_OK)) {
        intfft_plan_destroy(pl);
        return rc;
    }
    // USE_FLY = 0 (the bypass mux of int_fftNk.vhd:260-277, int_ifftNk.vhd:259-276): the butterflies are out of the path and only the
    // commutators move data.  In place terms nothing moves at all: position p of the core holds input index p and is read out as output index
    // bitrev(p), so ONE core is the bit reversal of the logical index (a pair: the identity), applied to the input wrapped to DATA_WIDTH bits in
    // the output container (k_convert), and a plan is that conversion followed by one bit permutation of the memory index -- none at all for the
    // cores' own beat orders (NATURAL / HALVES in -> BITREV out and the inverse's mirror move no data).  Results within 64 bits.
    if (!l1 && p->use_fly == 0 && pl->out_cb <= 8 && !diag_env("INTFFT_GENERIC_ONLY") && !diag_env("INTFFT_NO_BYPASS_COPY")) {
        bool ident = true;
        for (int j = 0; j < pl->L; ++j) { // output logical bit j = input logical bit L-1-j (single core) / j (pair)
            const int jin = p->direction == INTFFT_PAIR ? j : pl->L - 1 - j;
            pl->bypass_perm[order_mem_bit(p->out_order, pl->L, j)] = order_mem_bit(p->in_order, pl->L, jin);
        }
        for (int b = 0; b < pl->L; ++b) ident = ident && pl->bypass_perm[b] == b;
        if (!ident) {
            const size_t frame_bytes = ((size_t)2 << pl->L) * (size_t)pl->out_cb;
            size_t mb = 256;
            if (const char *e = diag_env("INTFFT_SCRATCH_MB")) mb = atoi(e) > 0 ? (size_t)atoi(e) : mb;
            pl->pair_frames = std::max<size_t>(1, (mb << 20) / frame_bytes);
            if (hipMalloc(&pl->pair_buf, pl->pair_frames * frame_bytes) != hipSuccess) {
                intfft_plan_destroy(pl);
                return INTFFT_ERR_ALLOC;
            }
            pl->scratch_frame_bytes = frame_bytes;
            pl->scratch_bytes = pl->pair_frames * frame_bytes;
        }
        (void)hipFree(pl->d_tw);
        pl->d_tw = nullptr;
        pl->lanes_mode = 3;
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), ident ? "bypass[k_convert]" : "bypass[k_convert|k_reorder]");
        *out = pl;
        return INTFFT_OK;
    }
    // BITREV_LANES (the serial stream of outbuf_half_path.vhd:160-172 / int_bitrev_order.vhd:82-104) at one end of a plan whose BITREV twin
    // has dedicated kernels: that twin + one bit permutation (BITREV <-> BITREV_LANES is a rotation of the memory index by one bit) through a
    // chunked middle buffer.  The packed 16-bit kernels of N = 128 .. 16384 carry the order as a store / load map of their own and never come here.
    if (!l1 && p->use_fly == 1 && (p->in_order == INTFFT_ORDER_BITREV_LANES) != (p->out_order == INTFFT_ORDER_BITREV_LANES) &&
        !diag_env("INTFFT_GENERIC_ONLY") && !diag_env("INTFFT_NO_LANES_COMPOSITE") &&
        !fast1024_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->use_fly, p->in_order, p->out_order) &&
        !fast1024x_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->use_fly, p->in_order, p->out_order) &&
        !fast4096_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->use_fly, p->in_order, p->out_order) &&
        !(fast16k_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->use_fly, p->in_order, p->out_order) &&
          fast16k_tables_ok(p->log2n, pl->h_tw.data(), p->twdl_width))) {
        const bool in_l = p->in_order == INTFFT_ORDER_BITREV_LANES;
        const int cb = in_l ? pl->in_cb : pl->out_cb;
        intfft_params q = *p;
        (in_l ? q.in_order : q.out_order) = INTFFT_ORDER_BITREV;
        auto dedicated = [](const intfft_plan *s) { return s && (s->passes.empty() || s->big20 || s->bigw || s->wide16 || s->is_pair); };
        if (cb <= 8 && create_plan(&pl->pair_f, &q, 0, hip_device) == INTFFT_OK && dedicated(pl->pair_f)) {
            const size_t frame_bytes = ((size_t)2 << pl->L) * (size_t)cb;
            size_t mb = 256;
            if (const char *e = diag_env("INTFFT_SCRATCH_MB")) mb = atoi(e) > 0 ? (size_t)atoi(e) : mb;
            pl->pair_frames = std::max<size_t>(1, (mb << 20) / frame_bytes);
            if (hipMalloc(&pl->pair_buf, pl->pair_frames * frame_bytes) != hipSuccess) {
                intfft_plan_destroy(pl);
                return INTFFT_ERR_ALLOC;
            }
            (void)hipFree(pl->d_tw); // the twin carries its own tables; the host copy stays for intfft_twiddles
            pl->d_tw = nullptr;
            pl->lanes_mode = in_l ? 2 : 1;
            pl->scratch_frame_bytes = frame_bytes;
            pl->scratch_bytes = pl->pair_frames * frame_bytes + pl->pair_f->scratch_bytes;
            std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), in_l ? "lanes[k_rotate1|%.40s]" : "lanes[%.40s|k_rotate1]", pl->pair_f->kernel_name);
            *out = pl;
            return INTFFT_OK;
        }
        if (pl->pair_f) intfft_plan_destroy(pl->pair_f);
        pl->pair_f = nullptr;
    }

    if (l1 && !diag_env("INTFFT_2D_GENERIC")) {
        // composite 2-D plan: the cores are 1-D sub-plans (NATURAL -> NATURAL) on re-laid-out data
        const int l2 = p->log2n - l1, F = p->format;
        auto sub = [&](int log2n, int dw, int direction, intfft_plan **o) {
            intfft_params q = *p;
            q.log2n = log2n, q.data_width = dw, q.direction = direction;
            q.in_order = q.out_order = INTFFT_ORDER_NATURAL;
            return create_plan(o, &q, 0, hip_device);
        };
        std::vector<StageDesc> st;
        int dw = p->data_width;
        rc = INTFFT_OK;
        if (p->direction != INTFFT_INV) {
            if (rc == INTFFT_OK) rc = sub(l1, dw, INTFFT_FWD, &pl->sub_col_f);
            if (rc == INTFFT_OK) rc = sub(l2, dw + F * l1, INTFFT_FWD, &pl->sub_row_f);
            if (rc == INTFFT_OK) rc = core_stages_2d(*p, l1, dw, false, st);
            for (const StageDesc &d : st)
                if (d.kind == KIND_TWMUL) pl->tw_f = d;
            dw += F * p->log2n;
        }
        if (p->direction != INTFFT_FWD) {
            st.clear();
            if (rc == INTFFT_OK) rc = sub(l2, dw, INTFFT_INV, &pl->sub_row_i);
            if (rc == INTFFT_OK) rc = sub(l1, dw + F * l2, INTFFT_INV, &pl->sub_col_i);
            if (rc == INTFFT_OK) rc = core_stages_2d(*p, l1, dw, true, st);
            for (const StageDesc &d : st)
                if (d.kind == KIND_TWMULC) pl->tw_i = d;
        }
        if (rc == INTFFT_OK) {
            const size_t frame_bytes = ((size_t)2 << pl->L) * (size_t)pl->out_cb;
            size_t layout_mb = 256; // per layout buffer; INTFFT_SCRATCH_MB bounds these as it bounds the 1-D plans' scratch
            if (const char *e = diag_env("INTFFT_SCRATCH_MB")) layout_mb = atoi(e) > 0 ? (size_t)atoi(e) : layout_mb;
            pl->buf2d_frames = std::max<size_t>(1, (layout_mb << 20) / frame_bytes);
            if (const char *e = diag_env("INTFFT_2D_CHUNK_FRAMES")) // diagnostics: exercise the chunk loop on small batches
                if (atoi(e) > 0) pl->buf2d_frames = std::min(pl->buf2d_frames, (size_t)atoi(e));
            for (int i = 0; i < 2 && rc == INTFFT_OK; ++i) rc = (int)hipMalloc(&pl->buf2d[i], pl->buf2d_frames * frame_bytes);
            pl->is2d = true;
            pl->n2d_bufs = 2;
        }
        if (rc != INTFFT_OK) {
            intfft_plan_destroy(pl);
            return rc;
        }
        const intfft_plan *a = pl->sub_col_f ? pl->sub_col_f : pl->sub_row_i, *b = pl->sub_row_f ? pl->sub_row_f : pl->sub_col_i;
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[%.24s|%.24s]", a->kernel_name, b->kernel_name);
        // 1024 x 1024, packed 16-bit forward: both cores and the multiplier in two launches on the tiles of the N = 2^20 two-pass plan
        pl->fused2d = fused2d_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->in_order, p->out_order);
        if (!(pl->tw_f.mw == 16 && pl->tw_f.sh_a == 0 && pl->tw_f.sh_b == p->twdl_width - 1 && pl->sub_col_f && pl->sub_row_f &&
              big2x_tables_ok(pl->fused2d == 8 ? 11 : 10, pl->sub_col_f->h_tw.data(), p->twdl_width)))
            pl->fused2d = 0;
        // ... and the inverse (4): the row cores as pass QB, the conj multiplier + the column cores on pass QA's tiles (k_big2x_ci)
        const int inv2d = fused2d_inv_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->in_order, p->out_order);
        if (!pl->fused2d && inv2d && pl->tw_i.mw == 16 && pl->tw_i.sh_a == 0 && pl->tw_i.sh_b == p->twdl_width - 1 && pl->sub_row_i && pl->sub_col_i &&
            big2x_tables_ok(inv2d == 4 ? 11 : 10, pl->sub_col_i->h_tw.data(), p->twdl_width) &&
            (inv2d == 3 || big2x_tables_ok(l2, pl->sub_row_i->h_tw.data(), p->twdl_width)))
            pl->fused2d = inv2d == 2 ? 6 : inv2d == 3 ? 9 : inv2d == 4 ? 10 : 4; // 6 (round 5): N = 2^21, the 2048-point row cores in k_rows2k_qtr; 9: three launches
        // ... and the pair (5): the forward two launches into the second layout buffer, the inverse two launches from there
        if (!pl->fused2d && p->direction == INTFFT_PAIR && !diag_env("INTFFT_2D_NO_FUSED_CORES") && pl->sub_col_f && pl->sub_row_f && pl->sub_row_i && pl->sub_col_i &&
            fused2d_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, INTFFT_FWD, p->in_order, INTFFT_ORDER_NATURAL) == 2 &&
            fused2d_inv_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, INTFFT_INV, INTFFT_ORDER_NATURAL, p->out_order) &&
            pl->tw_f.mw == 16 && pl->tw_f.sh_a == 0 && pl->tw_f.sh_b == p->twdl_width - 1 && pl->tw_i.mw == 16 && pl->tw_i.sh_a == 0 &&
            pl->tw_i.sh_b == p->twdl_width - 1 && big2x_tables_ok(10, pl->sub_col_f->h_tw.data(), p->twdl_width))
            pl->fused2d = 5;
        // ... and the pair at N = 2^21 (7, round 5): k_big2x_c<11> + k_rows2k_tr into the second layout buffer, k_rows2k_qtr + k_big2x_ci<., 11> from there
        if (!pl->fused2d && p->direction == INTFFT_PAIR && !diag_env("INTFFT_2D_NO_FUSED_CORES") && !diag_env("INTFFT_2D_NO_ROWS2K") && pl->sub_col_f &&
            pl->sub_row_f && pl->sub_row_i && pl->sub_col_i &&
            fused2d_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, INTFFT_FWD, p->in_order, INTFFT_ORDER_NATURAL) == 3 && l2 == 11 &&
            fused2d_inv_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, INTFFT_INV, INTFFT_ORDER_NATURAL, p->out_order) == 2 &&
            pl->tw_f.mw == 16 && pl->tw_f.sh_a == 0 && pl->tw_f.sh_b == p->twdl_width - 1 && pl->tw_i.mw == 16 && pl->tw_i.sh_a == 0 &&
            pl->tw_i.sh_b == p->twdl_width - 1 && big2x_tables_ok(10, pl->sub_col_f->h_tw.data(), p->twdl_width) &&
            big2x_tables_ok(11, pl->sub_row_f->h_tw.data(), p->twdl_width))
            pl->fused2d = 7;
        // ... and the pair at N = 2^22 = 2048 x 2048 (11): form 8's two launches into the second layout buffer, form 10's two from there
        if (!pl->fused2d && p->direction == INTFFT_PAIR && pl->sub_col_f && pl->sub_row_f && pl->sub_row_i && pl->sub_col_i &&
            fused2d_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, INTFFT_FWD, p->in_order, INTFFT_ORDER_NATURAL) == 8 &&
            fused2d_inv_supported(p->log2n, l1, p->data_width, p->twdl_width, p->format, p->rndmode, INTFFT_INV, INTFFT_ORDER_NATURAL, p->out_order) == 4 &&
            pl->tw_f.mw == 16 && pl->tw_f.sh_a == 0 && pl->tw_f.sh_b == p->twdl_width - 1 && pl->tw_i.mw == 16 && pl->tw_i.sh_a == 0 &&
            pl->tw_i.sh_b == p->twdl_width - 1 && big2x_tables_ok(11, pl->sub_col_f->h_tw.data(), p->twdl_width))
            pl->fused2d = 11;
        const intfft_plan *core1k = pl->fused2d == 4 ? pl->sub_row_i : (pl->fused2d == 6 || pl->fused2d == 9 || pl->fused2d == 10) ? pl->sub_col_i : pl->sub_col_f; // a 1024-point core of the plan (its twiddle tables)
        if (pl->fused2d) {
            // the fused launches run none of the 1-D sub-plans except form 3's rows, and forms 2 / 4 need one layout buffer only:
            // keep the core whose twiddle tables the tile kernels read (core1k), release the rest
            auto drop = [&](intfft_plan **sp) {
                if (*sp && *sp != core1k && !((pl->fused2d == 3 || pl->fused2d == 7) && *sp == pl->sub_row_f) && !((pl->fused2d == 6 || pl->fused2d == 9) && *sp == pl->sub_row_i)) {
                    intfft_plan_destroy(*sp);
                    *sp = nullptr;
                }
            };
            drop(&pl->sub_col_f), drop(&pl->sub_row_f), drop(&pl->sub_row_i), drop(&pl->sub_col_i);
            if (pl->fused2d == 2 || pl->fused2d == 4 || pl->fused2d == 6 || pl->fused2d == 8 || pl->fused2d == 10) {
                (void)hipFree(pl->buf2d[1]);
                pl->buf2d[1] = nullptr;
                pl->n2d_bufs = 1;
            }
            // the 1024-point cores' twiddles in the packed operand forms (the single-kernel sub-plans pack theirs on the fly)
            const size_t total = ((size_t)1 << 10) - 1;
            hipError_t e = hipMalloc((void **)&pl->d_tw16f, (total + 1) * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw16i, (total + 1) * sizeof(uint2));
            if (e == hipSuccess) e = launch_pack_twiddles16(core1k->d_tw, total, pl->d_tw16f, pl->d_tw16i, nullptr);
            if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw2d_tiles, ((size_t)1 << pl->L) * sizeof(uint32_t));
            if (e == hipSuccess) e = build_fused2d_table(pl->d_tw2d_tiles, pl->L, p->twdl_width, nullptr, l1);
            if (e == hipSuccess && pl->fused2d == 3 && l2 == 11 && p->out_order == INTFFT_ORDER_NATURAL && !diag_env("INTFFT_2D_NO_ROWS2K") &&
                big2x_tables_ok(11, pl->sub_row_f->h_tw.data(), p->twdl_width)) {
                // N = 2^21: the row cores and the store of X[k1 + 1024 k2] in ONE launch (k_rows2k_tr): two launches instead of three
                const size_t tot = ((size_t)1 << 11) - 1;
                e = hipMalloc((void **)&pl->d_tw16r, (tot + 1) * sizeof(uint2));
                if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw16ri, (tot + 1) * sizeof(uint2));
                if (e == hipSuccess) e = launch_pack_twiddles16(pl->sub_row_f->d_tw, tot, pl->d_tw16r, pl->d_tw16ri, nullptr);
                if (e == hipSuccess) { // the two-launch form writes X from the first layout buffer: the second one is never touched
                    (void)hipFree(pl->buf2d[1]);
                    pl->buf2d[1] = nullptr;
                    pl->n2d_bufs = 1;
                }
            }
            if (e == hipSuccess && (pl->fused2d == 8 || pl->fused2d == 10 || pl->fused2d == 11)) { // the 2048-point cores' packed table (columns and rows share it)
                const size_t tot = ((size_t)1 << 11) - 1;
                e = hipMalloc((void **)&pl->d_tw16r, (tot + 1) * sizeof(uint2));
                if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw16ri, (tot + 1) * sizeof(uint2));
                if (e == hipSuccess) e = launch_pack_twiddles16(core1k->d_tw, tot, pl->d_tw16r, pl->d_tw16ri, nullptr);
            }
            if (e == hipSuccess && (pl->fused2d == 6 || pl->fused2d == 7)) { // the 2048-point row core's packed table for k_rows2k_qtr (the pair: both row kernels)
                const size_t tot = ((size_t)1 << 11) - 1;
                e = hipMalloc((void **)&pl->d_tw16r, (tot + 1) * sizeof(uint2));
                if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw16ri, (tot + 1) * sizeof(uint2));
                if (e == hipSuccess) e = launch_pack_twiddles16((pl->fused2d == 7 ? pl->sub_row_f : pl->sub_row_i)->d_tw, tot, pl->d_tw16r, pl->d_tw16ri, nullptr);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) {
                intfft_plan_destroy(pl);
                return (int)e;
            }
            if (e == hipSuccess && !diag_env("INTFFT_ONE_STREAM")) {
                pl->wants_side = true;
                ExecCtx c; // the first context of the pool up front: no stream is created inside the first exec
                if (!ctx_acquire(pl, c)) {
                    intfft_plan_destroy(pl);
                    return (int)hipErrorOutOfMemory;
                }
                ctx_release(pl, c);
            }
            if (pl->fused2d == 2) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fused2d_kernel_name());
            else if (pl->fused2d == 4) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_big2x_qb|k_big2x_ci]");
            else if (pl->fused2d == 6) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_rows2k_qtr|k_big2x_ci]");
            else if (pl->fused2d == 8) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_cols2k_c|k_rows2k_tr]");
            else if (pl->fused2d == 10) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_rows2k_qtr|k_cols2k_ci]");
            else if (pl->fused2d == 11) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_cols2k_c|k_rows2k_tr|k_rows2k_qtr|k_cols2k_ci]");
            else if (pl->fused2d == 9) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[%.24s|k_big2x_ci]", pl->sub_row_i->kernel_name);
            else if (pl->fused2d == 7) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_big2x_c|k_rows2k_tr|k_rows2k_qtr|k_big2x_ci]");
            else if (pl->fused2d == 5) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_big2x_c|k_big2x_b|k_big2x_qb|k_big2x_ci]");
            else if (pl->d_tw16r) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_big2x_c|k_rows2k_tr]");
            else std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "2d[k_big2x_c|%.24s]", pl->sub_row_f->kernel_name);
        }
        *out = pl;
        return INTFFT_OK;
    }
    // INTFFT_GENERIC_ONLY=1 (diagnostics / A-B parity): plan with the generic LDS pass kernels only.
    // 2-D scheme plans in their flat form (INTFFT_2D_GENERIC=1: A/B parity of the composite form) also run on the generic
    // kernels: their column stages index the twiddle tables differently.
    const bool generic_only = diag_env("INTFFT_GENERIC_ONLY") != nullptr || l1 != 0;
    pl->fastsmall = !generic_only && fastsmall_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction,
                                                         p->use_fly, p->in_order, p->out_order);
    pl->fast1024 = !generic_only && fast1024_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode,
                                                       p->direction, p->use_fly, p->in_order, p->out_order);
    pl->fast4096 = !generic_only && !pl->fast1024 && fast4096_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode,
                                                       p->direction, p->use_fly, p->in_order, p->out_order);
    pl->fast16k = !generic_only && fast16k_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->use_fly, p->in_order,
                                                    p->out_order) &&
                  fast16k_tables_ok(p->log2n, pl->h_tw.data(), p->twdl_width);
    pl->fast1024x = !generic_only && fast1024x_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction,
                                        p->use_fly, p->in_order, p->out_order);
    pl->fast1024u = !generic_only && fast1024u_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly,
                                        p->in_order, p->out_order) &&
                    !diag_env("INTFFT_NO_FAST1024U");
    pl->fast1024ux = !generic_only && fast1024ux_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly,
                                          p->in_order, p->out_order) &&
                     !diag_env("INTFFT_NO_FAST1024U");
    if (pl->fast1024ux) { // per-stage multiplier regimes of the inverse core
        std::vector<StageDesc> st;
        const int dw_inv = p->direction == INTFFT_PAIR ? p->data_width + p->log2n : p->data_width;
        if (core_stages(*p, dw_inv, true, st) != INTFFT_OK || (int)st.size() != p->log2n) pl->fast1024ux = false;
        for (size_t i = 0; i < st.size() && pl->fast1024ux; ++i) {
            const StageDesc &d = st[i];
            if (d.s != (int)i || d.mw > 31 || d.sh_a + d.sh_b > 31) pl->fast1024ux = false;
            if (p->direction == INTFFT_INV && d.s >= 2 && d.sh_a != 0) pl->fast1024ux = false; // chained form
            pl->uxargs.st[i] = UxStage{d.sh_a + d.sh_b, ~((1u << d.sh_a) - 1u), d.mw};
        }
    }
    pl->fastw32 = !generic_only && !pl->fast1024 && !pl->fast1024u && !pl->fast1024ux && !pl->fast1024x &&
                  fastw32_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order,
                                    p->out_order) &&
                  !diag_env("INTFFT_NO_FASTW32");
    pl->fast4096w = !generic_only && !pl->fast4096 &&
                    fast4096w_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly,
                                        p->in_order, p->out_order) &&
                    !diag_env("INTFFT_NO_FASTW32");
    pl->w32inv = !generic_only && !pl->fast1024x && !pl->fast4096 && !pl->fast1024ux &&
                 w32inv_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order,
                                  p->out_order) &&
                 !diag_env("INTFFT_NO_FASTW32");
    const bool bigw_long = bigw_long_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order, p->out_order) &&
                           !diag_env("INTFFT_NO_BIGWLONG"); // N = 2^17 .. 2^20 (round 5): a pre-pass in front of the two passes
    pl->bigw = !generic_only &&
               !big20_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction, p->use_fly,
                                p->in_order, p->out_order) && // the packed three-pass kernels are faster where they apply
               (bigw_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order,
                               p->out_order) || bigw_long) &&
               !diag_env("INTFFT_NO_FASTW32");
    if (pl->fastw32 || pl->fast4096w || pl->w32inv || pl->bigw) {
        std::vector<StageDesc> st;
        if (core_stages(*p, p->data_width, p->direction == INTFFT_INV, st) != INTFFT_OK || (int)st.size() != p->log2n)
            pl->fastw32 = pl->fast4096w = pl->w32inv = pl->bigw = false;
        for (size_t i = 0; i < st.size() && (pl->fastw32 || pl->fast4096w || pl->w32inv || pl->bigw); ++i) {
            const StageDesc &d = st[i];
            // stages 1 and 0 may exceed 32 bits (by the 33rd / 34th bit) in unscaled forward plans: 64-bit tail
            bool tail = d.s <= 1 && p->format == 1 && p->direction == INTFFT_FWD && d.dtw <= 33 && d.wo <= 34;
            if (tail && d.wo > 32 && !pl->w32args.out64) pl->w32args.out64 = 1;
            // N = 2048 / 4096 block kernel: the whole last register round (STAGE 3..0) may run in 64 bits (out64 = 2) as long as
            // STAGE 4 still fits 32: 35 / 36-bit results (24-bit unscaled data at these lengths)
            const bool tail4 = pl->fast4096w && d.s <= 3 && p->format == 1 && p->direction == INTFFT_FWD &&
                               p->data_width + p->log2n - 4 <= 32 && d.wo <= 40 && d.mw + p->twdl_width <= 63;
            if (tail4 && !tail && (d.wo > 32 || d.dtw > 32)) {
                pl->w32args.out64 = 2;
                tail = true;
            }
            if (d.s < 0 || d.s > 19 || (!tail && (d.dtw > 32 || d.wo > 32 || d.mw > 32)) || d.sh_a + d.sh_b > 31 ||
                d.mw + p->twdl_width > (tail ? 63 : 62)) {
                pl->fastw32 = pl->fast4096w = pl->w32inv = pl->bigw = false;
                break;
            }
            pl->w32args.st[d.s] = W32Stage{d.sh_a + d.sh_b, ~((1u << d.sh_a) - 1u), 32 - d.mw, 32 - d.wo};
            if (d.s >= 2 && d.sh_a != 0) pl->w32args.masked = 1;
        }
        pl->w32args.inverse = p->direction == INTFFT_INV;
        pl->w32args.in16 = pl->in_cb == 2;
        pl->w32args.out16 = pl->out_cb == 2;
        pl->w32args.in_sh = 32 - p->data_width;
        if (pl->in_cb > 4 || (pl->out_cb > 4) != (pl->w32args.out64 != 0)) pl->fastw32 = pl->fast4096w = pl->w32inv = pl->bigw = false;
        if (pl->w32args.out64) pl->w32inv = pl->bigw = false; // 64-bit tail: forward wave / block kernels only
        if (pl->w32args.out64 == 2) pl->fastw32 = false;        // the 64-bit last round: the block kernel only
        pl->w32args.two_pass = pl->bigw && !diag_env("INTFFT_NO_TWOPASS");
        if (pl->bigw && p->log2n > 16) pl->w32args.two_pass = 2; // the long frames: always the pre-pass + the two passes
        if (pl->bigw) { // the cores' own orders: on the two-pass kernels only (round 5)
            pl->w32args.native = p->direction == INTFFT_INV ? ((p->out_order == INTFFT_ORDER_HALVES ? 1 : 0) | (p->in_order == INTFFT_ORDER_BITREV ? 2 : 0))
                                                             : ((p->in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (p->out_order == INTFFT_ORDER_BITREV ? 2 : 0));
            if (pl->w32args.native && !pl->w32args.two_pass) pl->bigw = false;
        }
    }
    pl->fastw64 = !generic_only && !pl->fastw32 && !pl->fast1024 && !pl->fast1024u && !pl->fast1024ux && !pl->fast1024x && !pl->fastsmall &&
                  pl->word == 8 && pl->in_cb >= 4 && pl->out_cb == 8 &&
                  fastw64_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order, p->out_order);
    if (pl->fastw64) {
        std::vector<StageDesc> st;
        if (core_stages(*p, p->data_width, p->direction == INTFFT_INV, st) != INTFFT_OK || (int)st.size() != p->log2n) pl->fastw64 = false;
        for (size_t i = 0; i < st.size() && pl->fastw64; ++i) {
            if (st[i].s < 0 || st[i].s > 9 || st[i].wo > 64 || st[i].dtw > 64) pl->fastw64 = false;
            else pl->st64[st[i].s] = st[i];
        }
        if (pl->fastw64 && !fastw64_plan_ok(p->log2n, pl->st64, p->format ? RND_UNSCALED : p->rndmode ? RND_ROUND : RND_TRUNC)) pl->fastw64 = false;
    }
    pl->fastw64b = !generic_only && !pl->fast4096w && !pl->w32inv && !pl->fast4096 && pl->word == 8 && pl->in_cb >= 4 && pl->out_cb == 8 &&
                   fastw64b_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order, p->out_order);
    if (pl->fastw64b) {
        std::vector<StageDesc> st;
        if (core_stages(*p, p->data_width, p->direction == INTFFT_INV, st) != INTFFT_OK || (int)st.size() != p->log2n) pl->fastw64b = false;
        for (size_t i = 0; i < st.size() && pl->fastw64b; ++i) {
            if (st[i].s < 0 || st[i].s > 11 || st[i].wo > 64 || st[i].dtw > 64) pl->fastw64b = false;
            else pl->st64[st[i].s] = st[i];
        }
        if (pl->fastw64b && !fastw64b_plan_ok(p->log2n, pl->st64, p->format ? RND_UNSCALED : p->rndmode ? RND_ROUND : RND_TRUNC)) pl->fastw64b = false;
    }
    if (pl->fastsmall) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fastsmall_kernel_name());
    } else if (pl->fast16k) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast16k_kernel_name());
        const size_t total = ((size_t)1 << pl->L) - 1; // the packed dot-product operand forms of the whole table
        hipError_t e = hipMalloc((void **)&pl->d_tw16f, (total + 1) * sizeof(uint2));
        if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw16i, (total + 1) * sizeof(uint2));
        if (e == hipSuccess) e = launch_pack_twiddles16(pl->d_tw, total, pl->d_tw16f, pl->d_tw16i, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            intfft_plan_destroy(pl);
            return (int)e;
        }
    } else if (pl->fastw64b) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fastw64b_kernel_name(p->direction));
    } else if (pl->fastw64) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fastw64_kernel_name(p->direction));
    } else if (pl->w32inv) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", w32inv_kernel_name(p->log2n));
    } else if (pl->fast4096w) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast4096w_kernel_name());
    } else if (pl->fastw32) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fastw32_kernel_name());
    } else if (pl->fast1024ux) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast1024ux_kernel_name());
    } else if (pl->fast1024u) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast1024u_kernel_name());
    } else if (pl->fast4096) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast4096_kernel_name());
    } else if (pl->fast1024x) {
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast1024x_kernel_name());
    } else if (pl->fast1024) {
        pl->fargs.dw = p->data_width;
        pl->fargs.log2n = p->log2n;
        pl->fargs.twd = p->twdl_width;
        pl->fargs.rnd = p->rndmode ? RND_ROUND : RND_TRUNC;
        pl->fargs.out_bitrev = p->out_order == INTFFT_ORDER_BITREV ? 1 : p->out_order == INTFFT_ORDER_BITREV_LANES ? 2 : 0;
        pl->fargs.in_halves = p->in_order == INTFFT_ORDER_HALVES;
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", fast1024_kernel_name());
    } else {
        if ((rc = build_passes(*pl)) != INTFFT_OK) {
            intfft_plan_destroy(pl);
            return rc;
        }
        pl->big20 = !generic_only && big20_supported(p->log2n, p->data_width, p->twdl_width, p->format, p->rndmode, p->direction,
                                    p->use_fly, p->in_order, p->out_order) &&
                    !diag_env("INTFFT_NO_BIG20");
        // pairs without a pair kernel of their own whose two halves both have dedicated kernels: forward sub-plan, middle buffer in natural order,
        // inverse sub-plan (64-bit words since round 3; 32-bit words -- general widths within 32 bits -- since round 4)
        if (!generic_only && !l1 && p->direction == INTFFT_PAIR && (pl->word == 8 || pl->word == 4) && !pl->big20 && !pl->bigw && p->use_fly == 1 &&
            !diag_env("INTFFT_NO_PAIR_COMPOSITE")) {
            intfft_params qf = *p, qi = *p;
            qf.direction = INTFFT_FWD, qf.out_order = INTFFT_ORDER_NATURAL;
            qi.direction = INTFFT_INV, qi.in_order = INTFFT_ORDER_NATURAL, qi.data_width = p->data_width + (p->format ? p->log2n : 0);
            auto dedicated = [](const intfft_plan *q) { return q && (q->passes.empty() || q->big20 || q->bigw || q->wide16); };
            int rs = qi.data_width <= 64 ? create_plan(&pl->pair_f, &qf, 0, hip_device) : INTFFT_ERR_INVALID;
            if (rs == INTFFT_OK && dedicated(pl->pair_f)) rs = create_plan(&pl->pair_i, &qi, 0, hip_device);
            if (rs == INTFFT_OK && dedicated(pl->pair_f) && dedicated(pl->pair_i) && pl->pair_f->out_cb == pl->pair_i->in_cb) {
                const size_t frame_bytes = ((size_t)2 << pl->L) * (size_t)pl->pair_i->in_cb;
                size_t mb = 256;
                if (const char *e = diag_env("INTFFT_SCRATCH_MB")) mb = atoi(e) > 0 ? (size_t)atoi(e) : mb;
                pl->pair_frames = std::max<size_t>(1, (mb << 20) / frame_bytes);
                if (hipMalloc(&pl->pair_buf, pl->pair_frames * frame_bytes) != hipSuccess) {
                    intfft_plan_destroy(pl);
                    return INTFFT_ERR_ALLOC;
                }
                pl->is_pair = true;
                pl->scratch_frame_bytes = frame_bytes;
                pl->scratch_bytes = pl->pair_frames * frame_bytes + pl->pair_f->scratch_bytes + pl->pair_i->scratch_bytes;
                // the two sub-plans carry their own tables and passes: the parent's were only needed to decide eligibility
                (void)hipFree(pl->d_tw);
                pl->d_tw = nullptr;
                pl->passes.clear();
                pl->passes.shrink_to_fit();
                std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "pair[%.24s|%.24s]", pl->pair_f->kernel_name, pl->pair_i->kernel_name);
                *out = pl;
                return INTFFT_OK;
            }
            if (pl->pair_f) intfft_plan_destroy(pl->pair_f);
            if (pl->pair_i) intfft_plan_destroy(pl->pair_i);
            pl->pair_f = pl->pair_i = nullptr;
        }
        pl->big_two_pass = pl->big20 && p->log2n <= 16 && p->direction != INTFFT_PAIR && !diag_env("INTFFT_NO_TWOPASS");
        // N = 2^17, 2^18 forward / inverse: the 32-register pass of stages 8..L-1 (it shares twiddles by
        // quarter turns: verified on this plan's tables)
        // (round mode: the forward pass on its own instantiations since round 4 -- quarter turns through the negated twiddle, group4)
        const bool big2p = pl->big20 && (p->direction == INTFFT_INV || p->direction == INTFFT_FWD) && big2p_supported(p->log2n) &&
                           !diag_env("INTFFT_NO_TWOPASS") && big2p_tables_ok(p->log2n, pl->h_tw.data(), p->twdl_width);
        if (big2p) pl->big_two_pass = true;
        const bool big2p_pair = pl->big20 && !p->rndmode && p->direction == INTFFT_PAIR && big2p_supported(p->log2n) && !diag_env("INTFFT_NO_TWOPASS") &&
                                big2p_tables_ok(p->log2n, pl->h_tw.data(), p->twdl_width);
        pl->big_pair256 = pl->big20 && (p->log2n <= 16 || big2p_pair) && p->direction == INTFFT_PAIR && !diag_env("INTFFT_NO_TWOPASS");
        // N = 2^19, 2^20 forward (truncate mode, and since round 4 RNDMODE = 1 on its own instantiations), natural or BITREV order out: 1024 rows x
        // 1024 columns in two ten-stage passes (intfft_big2x.hip)
        const bool big2x = pl->big20 && p->direction == INTFFT_FWD && (p->out_order == INTFFT_ORDER_NATURAL || p->out_order == INTFFT_ORDER_BITREV) &&
                           big2x_supported(p->log2n) &&
                           !diag_env("INTFFT_NO_TWOPASS") && big2x_tables_ok(p->log2n, pl->h_tw.data(), p->twdl_width);
        if (big2x) pl->big_two_pass = true;
        // ... and the inverse from natural or BITREV order (natural or HALVES order out): k_big2x_qb / k_big2x_qa
        const bool big2x_inv = pl->big20 && p->direction == INTFFT_INV && (p->in_order == INTFFT_ORDER_NATURAL || p->in_order == INTFFT_ORDER_BITREV) &&
                               big2x_supported(p->log2n) &&
                               !diag_env("INTFFT_NO_TWOPASS") && big2x_tables_ok(p->log2n, pl->h_tw.data(), p->twdl_width);
        if (big2x_inv) pl->big_two_pass = true;
        // class 1: the first pass within int32 (17 .. 27-bit data); class 2 (round 5): DATA_WIDTH up to 32, the first pass on 64-bit words too
        const int wcls = generic_only ? 0 : wide16_class(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order, p->out_order);
        pl->wide16 = wcls != 0 && !diag_env("INTFFT_NO_WIDE16");
        pl->wargs.w64 = wcls == 2;
        const int lcls = (!generic_only && !pl->wide16 && !diag_env("INTFFT_NO_WIDELONG"))
                             ? widelong_class(p->log2n, p->data_width, p->twdl_width, p->format, p->direction, p->use_fly, p->in_order, p->out_order) : 0;
        pl->widelong = lcls != 0;
        if (pl->widelong) pl->wide16 = true, pl->wargs.w64 = lcls == 2; // the branches below fill the stage list of all NFFT stages
        pl->wargs.native = p->direction == INTFFT_INV ? ((p->out_order == INTFFT_ORDER_HALVES ? 1 : 0) | (p->in_order == INTFFT_ORDER_BITREV ? 2 : 0))
                                                       : ((p->in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (p->out_order == INTFFT_ORDER_BITREV ? 2 : 0));
        if (pl->wide16 && (wcls == 2 || lcls == 2)) { // every stage on the 64-bit butterflies (wfly64 / wdit64): exact 64-bit products, the slice inside one dword pair
            std::vector<StageDesc> st;
            const int LL = p->log2n;
            const bool inv = p->direction == INTFFT_INV;
            if (core_stages(*p, p->data_width, inv, st) != INTFFT_OK || (int)st.size() != LL) pl->wide16 = false;
            pl->wargs.dw = p->data_width;
            for (int i = 0; i < LL && pl->wide16; ++i) {
                const StageDesc &d = st[i];
                WideStage &w = pl->wargs.st[i]; // forward: processing order (STAGE LL - 1 - i); inverse: STAGE i
                const int wslice = inv ? d.mw : d.wo; // the multiplier's result width
                w.sh = d.sh_a + d.sh_b;
                w.keep = d.sh_a >= 32 ? 0u : ~((1u << d.sh_a) - 1u);
                w.az = d.sh_a == 0;
                w.s2 = w.s3 = 0;
                w.w32 = wslice - 32;
                if (pl->widelong && !inv && i < LL - 16) { // k_wide_pre (int32, forward): the slice of wo <= 32 bits
                    w.s3 = 32 - wslice;
                    if (w.s3 < 0 || w.s3 > 31) pl->wide16 = false;
                }
                if (d.s != (inv ? i : LL - 1 - i)) pl->wide16 = false;
                if (d.s < 2) continue; // multiplier-free
                if (d.sh_a >= 32 || w.sh > 31 || d.mw + p->twdl_width > 64 || wslice > 48 || w.sh + wslice > 64) pl->wide16 = false;
                if (w.w32 >= 1 && w.sh + w.w32 > 32) pl->wide16 = false;
            }
            if (p->data_width + LL > 48) pl->wide16 = false;
        } else if (pl->wide16 && p->direction == INTFFT_INV) { // the inverse: STAGE 0 .. 7 in pass 1 (int32), 8 .. LL-1 in pass 2 (64-bit); st[s] = STAGE s
            std::vector<StageDesc> st;
            const int LL = p->log2n;
            if (core_stages(*p, p->data_width, true, st) != INTFFT_OK || (int)st.size() != LL) pl->wide16 = false;
            pl->wargs.dw = p->data_width;
            for (int s = 0; s < LL && pl->wide16; ++s) {
                const StageDesc &d = st[s];
                WideStage &w = pl->wargs.st[s];
                w.sh = d.sh_a + d.sh_b;
                w.keep = ~((1u << d.sh_a) - 1u);
                w.az = d.sh_a == 0;
                w.s2 = w.sh + d.mw - 32; // the slice is the multiplier's own width mw = DATA_WIDTH + s
                w.s3 = 32 - d.mw;
                w.w32 = d.mw - 32;
                if (d.s != s || d.mw != p->data_width + s || d.mw + p->twdl_width > 64) pl->wide16 = false;
                if (s >= 2 && s < 8 && (w.s2 < 0 || w.s2 > 31 || w.s3 < 0)) pl->wide16 = false;
                if (s >= 8 && w.w32 >= 1 && w.sh + w.w32 > 32) pl->wide16 = false;
                if (s >= 8 && (d.mw > 40 || w.sh + d.mw > 64)) pl->wide16 = false;
            }
        } else if (pl->wide16) {
            std::vector<StageDesc> st;
            const int LL = p->log2n; // 13 .. 16: STAGE LL-1 .. 8 in pass 1 (int32), 7 .. 0 in pass 2 (64-bit)
            if (core_stages(*p, p->data_width, false, st) != INTFFT_OK || (int)st.size() != LL) pl->wide16 = false;
            pl->wargs.dw = p->data_width;
            pl->wargs.r32 = pl->widelong && !diag_env("INTFFT_NO_WIDELONG_R32");
            for (int ii = 0; ii < LL && pl->wide16; ++ii) {
                const StageDesc &d = st[ii];
                WideStage &w = pl->wargs.st[ii];
                w.sh = d.sh_a + d.sh_b;
                w.keep = ~((1u << d.sh_a) - 1u);
                w.az = d.sh_a == 0;
                w.s2 = w.sh + d.wo - 32;
                w.s3 = 32 - d.wo;
                w.w32 = d.wo - 32;
                if (d.s != LL - 1 - ii || d.mw + p->twdl_width > 64) pl->wide16 = false;
                if (d.s >= 8 && d.s < 16 && (w.s2 < 0 || w.s2 > 31 || w.s3 < 0)) pl->wide16 = false;
                if (d.s >= 16 && (w.sh < 0 || w.sh > 31 || w.s3 < 0 || w.s3 > 31)) pl->wide16 = false; // k_wide_pre: the general slice form
                // pass 2: widths beyond 32 use v_alignbit + v_bfe (slice inside one dword pair), the others the general form
                if (d.s >= 4 && d.s < 8 && (d.wo > 32 || w.sh + d.wo - 32 < 0 || w.sh + d.wo - 32 > 31)) pl->wargs.r32 = 0; // round 1 of pass 2 on int32 (long frames)
                if (d.s >= 2 && d.s < 8 && w.w32 >= 1 && w.sh + w.w32 > 32) pl->wide16 = false;
                if (d.s < 8 && (d.wo > 40 || w.sh + d.wo > 64)) pl->wide16 = false;
            }
        }
        if (!pl->wide16) pl->widelong = false;
        std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s",
                      pl->bigw ? bigw_kernel_name(p->direction, pl->w32args.two_pass) : pl->widelong ? (p->direction == INTFFT_INV ? (pl->wargs.w64 ? "k_wide64_q1+k_wide16_q2+k_wide_post" : "k_wide16_q1+q2+k_wide_post") : pl->wargs.w64 ? "k_wide_pre+k_wide64_p1+k_wide16_p2" : "k_wide_pre+k_wide16_p1+p2") : pl->wide16 ? wide16_kernel_name(p->direction, pl->wargs.w64) : big2x ? big2x_kernel_name() : big2x_inv ? "k_big2x_qb/k_big2x_qa" : pl->big20 ? big20_kernel_name(p->direction, (big2p || big2p_pair) ? 2 : (pl->big_two_pass || pl->big_pair256), (p->direction == INTFFT_INV ? p->in_order : p->out_order) == INTFFT_ORDER_BITREV) : pl->word == 2 ? pass16_kernel_name() : pass_kernel_name(pl->word));
        if (l1) std::snprintf(pl->kernel_name, sizeof(pl->kernel_name), "%s", pass_kernel_name(pl->word));
        // narrow data (DATA_WIDTH 9 .. 15) on the packed multi-pass kernels: int16 scratch words and the packed twiddle forms, as word == 2
        const bool narrow_big = pl->big20 && p->data_width != 16;
        if (pl->word == 2 || narrow_big) {
            const size_t total = ((size_t)1 << pl->L) - 1;
            hipError_t e = hipMalloc((void **)&pl->d_tw16f, (total + 1) * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc((void **)&pl->d_tw16i, (total + 1) * sizeof(uint2));
            if (e == hipSuccess) e = launch_pack_twiddles16(pl->d_tw, total, pl->d_tw16f, pl->d_tw16i, nullptr);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) {
                intfft_plan_destroy(pl);
                return (int)e;
            }
        }
        if (pl->passes.size() > 1 || pl->big20 || pl->bigw || pl->wide16) {
            // scratch words: int32 for the general-width three-pass kernels, else the (first) pass word
            const size_t frame_bytes = ((size_t)2 << pl->L) * (size_t)(pl->bigw ? 4 : (pl->widelong && pl->wargs.w64 && p->direction == INTFFT_INV) ? 16 : (pl->widelong && (pl->wargs.w64 || p->direction == INTFFT_INV)) ? 12 : (pl->wide16 && (pl->wargs.w64 || pl->widelong)) ? 8 : pl->wide16 ? 4 : (pl->word == 2 || narrow_big) ? 2 : pl->passes[0].word);
            // two scratch halves on two streams where the passes of a plan differ in what bounds them (the two-pass plans of N = 2^19 / 2^20:
            // a latency-bound column pass beside a bandwidth-bound row pass; the 24-bit-class kernels: +7 % and +8 %); the other
            // multi-pass families lose 1-9 % that way (measured) and keep one 256 MiB scratch on the caller's stream
            const bool dual = (big2x || big2x_inv || pl->wide16 || (diag_env("INTFFT_TWO_STREAMS") && (pl->big20 || pl->bigw))) && !diag_env("INTFFT_ONE_STREAM");
            size_t scratch_mb = 128; // per scratch buffer; two halves together are about the Infinity Cache
            if (!dual && (pl->big20 || pl->bigw || pl->wide16)) scratch_mb = 256;
            if (const char *e = diag_env("INTFFT_SCRATCH_MB")) scratch_mb = atoi(e) > 0 ? (size_t)atoi(e) : scratch_mb;
            pl->scratch_frames = std::max<size_t>(1, (scratch_mb << 20) / frame_bytes);
            if (pl->wide16 && pl->L < 16) { // a chunk = whole virtual frames (a partial one still occupies its whole region of the scratch)
                pl->scratch_gran = (size_t)1 << (16 - pl->L);
                pl->scratch_frames = std::max(pl->scratch_gran, pl->scratch_frames / pl->scratch_gran * pl->scratch_gran);
            }
            pl->scratch_frame_bytes = frame_bytes;
            pl->scratch_bytes = pl->scratch_frames * frame_bytes;
            hipError_t e = hipMalloc(&pl->d_scratch, pl->scratch_bytes);
            if (e == hipSuccess && dual) {
                e = hipMalloc(&pl->d_scratch2, pl->scratch_bytes);
                pl->dual_scratch = pl->wants_side = true;
                ExecCtx c;
                if (e == hipSuccess && !ctx_acquire(pl, c)) e = hipErrorOutOfMemory;
                if (e == hipSuccess) ctx_release(pl, c);
                if (e == hipSuccess) pl->scratch_bytes *= 2; // what intfft_plan_get_info reports: both halves
            }
            if (e != hipSuccess) {
                intfft_plan_destroy(pl);
                return (int)e;
            }
        }
    }
    *out = pl;
    return INTFFT_OK;
}

int intfft_plan_destroy(intfft_plan *plan)
{
    if (!plan) return INTFFT_ERR_NULL;
    {
        DeviceGuard guard(plan->device);
        if (plan->d_tw) (void)hipFree(plan->d_tw);
        if (plan->d_tw2d) (void)hipFree(plan->d_tw2d);
        if (plan->d_tw2d_tiles) (void)hipFree(plan->d_tw2d_tiles);
        if (plan->d_tw16r) (void)hipFree(plan->d_tw16r);
        if (plan->d_tw16ri) (void)hipFree(plan->d_tw16ri);
        for (void *b : plan->buf2d)
            if (b) (void)hipFree(b);
        for (intfft_plan *sp : {plan->sub_col_f, plan->sub_row_f, plan->sub_row_i, plan->sub_col_i, plan->pair_f, plan->pair_i})
            if (sp) intfft_plan_destroy(sp);
        if (plan->pair_buf) (void)hipFree(plan->pair_buf);
        if (plan->d_scratch) (void)hipFree(plan->d_scratch);
        if (plan->d_scratch2) (void)hipFree(plan->d_scratch2);
        for (ExecCtx &c : plan->ctx_pool) ctx_destroy(c);
        if (plan->shard_in) (void)hipFree(plan->shard_in);
        if (plan->shard_out) (void)hipFree(plan->shard_out);
        if (plan->rccl_owner == plan) rccl_release(plan);
        else rccl_leave(plan);
        for (hipStream_t st : {plan->s_shard, plan->s_shard2, plan->s_call})
            if (st) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : plan->shard_ev)
            if (ev) (void)hipEventDestroy(ev);
        free_stream_state(plan);
        if (plan->d_tw16f) (void)hipFree(plan->d_tw16f);
        if (plan->d_tw16i) (void)hipFree(plan->d_tw16i);
    }
    delete plan;
    return INTFFT_OK;
}

int intfft_plan_get_info(const intfft_plan *plan, intfft_plan_info *info)
{
    if (!plan || !info) return INTFFT_ERR_NULL;
    std::memset(info, 0, sizeof(*info));
    info->in_bits = plan->in_bits;
    info->out_bits = plan->out_bits;
    info->in_container = plan->in_cb;
    info->out_container = plan->out_cb;
    const bool fast = plan->fast1024 || plan->fast4096 || plan->fast16k || plan->fast1024x || plan->fast1024u || plan->fast1024ux || plan->fastw32 || plan->fast4096w || plan->w32inv || plan->fastsmall || plan->fastw64 || plan->fastw64b;
    if (plan->lanes_mode == 3) {
        info->n_passes = plan->pair_buf || plan->pair_frames ? 2 : 1;
        info->compute_word = plan->out_cb;
        info->fast_path = 1;
        info->scratch_bytes = plan->owns_scratch ? plan->scratch_bytes : 0;
        std::snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", plan->kernel_name);
        return INTFFT_OK;
    }
    if (plan->lanes_mode) {
        intfft_plan_info sf;
        if (intfft_plan_get_info(plan->pair_f, &sf) != INTFFT_OK) return INTFFT_ERR_INVALID;
        info->n_passes = sf.n_passes + 1;
        info->compute_word = sf.compute_word;
        info->fast_path = sf.fast_path;
        info->scratch_bytes = plan->owns_scratch ? plan->scratch_bytes : 0;
        std::snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", plan->kernel_name);
        return INTFFT_OK;
    }
    if (plan->is_pair) {
        intfft_plan_info sf, si;
        if (intfft_plan_get_info(plan->pair_f, &sf) != INTFFT_OK || intfft_plan_get_info(plan->pair_i, &si) != INTFFT_OK) return INTFFT_ERR_INVALID;
        info->n_passes = sf.n_passes + si.n_passes;
        info->compute_word = std::max(sf.compute_word, si.compute_word);
        info->fast_path = sf.fast_path && si.fast_path;
        info->scratch_bytes = plan->owns_scratch ? plan->scratch_bytes : 0;
        std::snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", plan->kernel_name);
        return INTFFT_OK;
    }
    if (plan->is2d) {
        intfft_plan_info si;
        int n = 0;
        for (const intfft_plan *sp : {plan->sub_col_f, plan->sub_row_f, plan->sub_row_i, plan->sub_col_i})
            if (sp && intfft_plan_get_info(sp, &si) == INTFFT_OK) n += si.n_passes;
        const int cores = (plan->sub_col_f ? 1 : 0) + (plan->sub_row_i ? 1 : 0);
        info->n_passes = n + 2 * cores + (cores == 2 ? 0 : 1); // per direction: layout change in or out + the middle one (multiplier fused in); a pair shares its middle
        if (plan->fused2d == 2 || plan->fused2d == 4 || plan->fused2d == 6 || plan->fused2d == 8 || plan->fused2d == 10) info->n_passes = 2;
        if (plan->fused2d == 5 || plan->fused2d == 7 || plan->fused2d == 11) info->n_passes = 4;
        if (plan->fused2d == 3 && intfft_plan_get_info(plan->sub_row_f, &si) == INTFFT_OK) info->n_passes = plan->d_tw16r ? 2 : 2 + si.n_passes;
        if (plan->fused2d == 9 && intfft_plan_get_info(plan->sub_row_i, &si) == INTFFT_OK) info->n_passes = 2 + si.n_passes;
        info->compute_word = plan->fused2d ? 2 : 0;
        info->fast_path = 0;
        // the whole device footprint of the plan beyond its twiddle tables: the two layout buffers AND the sub-plans' own scratch
        info->scratch_bytes = (size_t)plan->n2d_bufs * plan->buf2d_frames * ((size_t)2 << plan->L) * (size_t)plan->out_cb;
        for (const intfft_plan *sp : {plan->sub_col_f, plan->sub_row_f, plan->sub_row_i, plan->sub_col_i})
            if (sp) info->scratch_bytes += sp->scratch_bytes;
        if (!plan->owns_scratch) info->scratch_bytes = 0;
        std::snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", plan->kernel_name);
        return INTFFT_OK;
    }
    info->n_passes = fast ? 1 : (plan->big_two_pass || (plan->bigw && plan->w32args.two_pass == 1)) ? 2 : ((plan->big20 && !plan->wide16) || plan->bigw) ? 3 : plan->widelong ? 3 : plan->wide16 ? 2 : (int)plan->passes.size();
    info->compute_word = (plan->fastw64 || plan->fastw64b) ? 8 : (plan->fast1024u || plan->fast1024ux || plan->fastw32 || plan->fast4096w || plan->w32inv) ? 4 : (fast || (plan->big20 && !plan->wide16 && !plan->bigw)) ? 2 : plan->word;
    info->fast_path = fast ? 1 : 0;
    info->scratch_bytes = plan->owns_scratch ? plan->scratch_bytes : 0;
    std::snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", plan->kernel_name);
    return INTFFT_OK;
}

// The 2-D scheme as a sequence of launches.  Layouts are bit permutations of the frame index
// (launch_bitperm), the cores are 1-D sub-plans over N2 * frames (columns) / N1 * frames (rows) short frames:
//   forward  user (in_order) -> [n2][n1] | N1-point int_fftNk | [n2][k1] -> [k1][n2] | x W_N^(k1 n2) | N2-point int_fftNk
//            | [k1][k2] -> user (out_order, X[k1 + N1 k2])
//   inverse  user (in_order, X) -> [k1][k2] | N2-point int_ifftNk | x conj W | [k1][n2] -> [n2][k1] | N1-point int_ifftNk
//            | [n2][n1] -> user (out_order)          pair: forward up to [k1][k2], then the inverse from there
static int exec_core(intfft_plan *plan, const void *d_in, void *d_out, size_t batch, hipStream_t stream, char *ws);

// Frames per layout buffer of a 2-D plan for a call of `batch` frames on a caller-supplied workspace (plan-owned buffers hold
// buf2d_frames): the chunking below is the same in both cases.
static bool dual_2d(const intfft_plan *pl)
{
    // (the two-launch N = 2^21 plan stays on one stream: its row kernel is one 135 KiB workgroup per CU, which cannot share a CU with the column
    // pass's 68 KiB workgroups of the other chunk -- 269 Gsample/s on one stream against 255 on two)
    return pl->fused2d && pl->wants_side && pl->buf2d_frames / 2 >= 1 && (pl->fused2d != 3 || (pl->sub_row_f->scratch_bytes == 0 && !pl->d_tw16r)) &&
           (pl->fused2d != 9 || pl->sub_row_i->scratch_bytes == 0) && pl->fused2d != 6 && pl->fused2d != 7 && pl->fused2d != 8 && pl->fused2d != 10 && pl->fused2d != 11;
}
static size_t ws_frames_2d(const intfft_plan *pl, size_t batch)
{
    if (dual_2d(pl)) return batch <= pl->buf2d_frames / 2 ? batch : pl->buf2d_frames;
    return std::min(batch, pl->buf2d_frames);
}

// Workspace a call of `batch` frames needs when the caller supplies it (0: the plan is a single launch).  Monotone in `batch`.
static size_t ws_need(const intfft_plan *pl, size_t batch)
{
    if (batch == 0) return 0;
    if (pl->is2d) {
        const size_t bf = ws_frames_2d(pl, batch);
        size_t sub = 0;
        const int l1 = pl->l1, l2 = pl->L - pl->l1;
        if (pl->fused2d == 3) sub = ws_need(pl->sub_row_f, bf << l1);
        else if (pl->fused2d == 9) sub = ws_need(pl->sub_row_i, bf << l1);
        else if (!pl->fused2d) {
            if (pl->sub_col_f) sub = std::max(sub, ws_need(pl->sub_col_f, bf << l2));
            if (pl->sub_row_f) sub = std::max(sub, ws_need(pl->sub_row_f, bf << l1));
            if (pl->sub_row_i) sub = std::max(sub, ws_need(pl->sub_row_i, bf << l1));
            if (pl->sub_col_i) sub = std::max(sub, ws_need(pl->sub_col_i, bf << l2));
        }
        return (size_t)pl->n2d_bufs * ws_align(bf * ((size_t)2 << pl->L) * (size_t)pl->out_cb) + sub;
    }
    if (pl->lanes_mode == 3) return pl->pair_frames ? ws_align(std::min(pl->pair_frames, batch) * pl->scratch_frame_bytes) : 0;
    if (pl->lanes_mode) {
        const size_t pf = std::min(pl->pair_frames, batch);
        return ws_align(pf * pl->scratch_frame_bytes) + ws_need(pl->pair_f, pf);
    }
    if (pl->is_pair) {
        const size_t pf = std::min(pl->pair_frames, batch);
        return ws_align(pf * pl->scratch_frame_bytes) + std::max(ws_need(pl->pair_f, pf), ws_need(pl->pair_i, pf));
    }
    if (pl->scratch_frames == 0) return 0;
    if (batch <= pl->scratch_frames) return ws_align((batch + pl->scratch_gran - 1) / pl->scratch_gran * pl->scratch_gran * pl->scratch_frame_bytes);
    return (pl->dual_scratch ? 2 : 1) * ws_align(pl->scratch_frames * pl->scratch_frame_bytes);
}

static int exec_2d(intfft_plan *pl, const void *d_in, void *d_out, size_t batch, hipStream_t stream, char *ws)
{
    const int L = pl->L, l1 = pl->l1, l2 = L - l1;
    // the layout buffers and what the sub-plans may use: the plan's own, or carved from the caller's workspace
    const size_t bufs_frames = ws ? ws_frames_2d(pl, batch) : pl->buf2d_frames;
    const size_t buf_bytes = ws_align(bufs_frames * ((size_t)2 << L) * (size_t)pl->out_cb);
    void *const buf0 = ws ? ws : pl->buf2d[0];
    void *const buf1 = ws ? (pl->n2d_bufs == 2 ? ws + buf_bytes : nullptr) : pl->buf2d[1];
    char *const subws = ws ? ws + (size_t)pl->n2d_bufs * buf_bytes : nullptr;
    const intfft_params &p = pl->p;
    const size_t in_frame = ((size_t)2 << L) * (size_t)pl->in_cb, out_frame = ((size_t)2 << L) * (size_t)pl->out_cb;
    int perm[24];
    hipError_t e = hipSuccess;
    int rc = INTFFT_OK;
    const bool fuse = diag_env("INTFFT_2D_NO_FUSE") == nullptr; // diagnostics: the multiplier as its own launch (k_twmul)
    if (pl->fused2d) {
        // Two launches (N2 = 1024: column cores + multiplier, row cores + store, one layout buffer as the scratch) or three (N2 > 1024: the
        // column cores + multiplier on tiles, the row sub-plan, ONE layout change [rho][k2] -> X[brev10(rho) + 1024 k2]) per chunk.  The
        // chunks of a batch alternate between the caller's stream and the plan's side stream, each on its own half of the layout
        // buffers (section 4.2d: the strided column pass of one chunk beside the streaming launches of the other); under stream capture, and
        // when the row sub-plan owns a scratch of its own, everything stays on the caller's stream.
        const size_t half = pl->buf2d_frames / 2;
        SideStream side(pl, stream);
        if (dual_2d(pl) && batch > half && (e = side.fork()) != hipSuccess) return (int)e;
        const bool dual = side.on;
        const size_t chunk = dual ? half : bufs_frames;
        size_t ci = 0;
        for (size_t f = 0; f < batch && e == hipSuccess && rc == INTFFT_OK; f += chunk, ++ci) {
            const size_t nf = std::min(chunk, batch - f);
            const bool odd = dual && (ci & 1);
            hipStream_t st = odd ? side.c.side : stream;
            const size_t off = odd ? half * out_frame : 0; // (layout buffers are sized in frames of the output container)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(static_cast<const char *>(d_in) + f * in_frame);
            char *dst = static_cast<char *>(d_out) + f * out_frame;
            uint32_t *b0 = reinterpret_cast<uint32_t *>(static_cast<char *>(buf0) + off);
            if (pl->fused2d == 5) { // the pair: X in natural order in the second layout buffer between the two directions
                uint32_t *b1 = reinterpret_cast<uint32_t *>(static_cast<char *>(buf1) + off);
                e = launch_fused2d(p.twdl_width, src, b1, b0, pl->d_tw16f, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf, p.in_order == INTFFT_ORDER_HALVES, st);
                if (e == hipSuccess)
                    e = launch_fused2d_inv(p.twdl_width, b1, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16f, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf,
                                           p.out_order == INTFFT_ORDER_HALVES, st);
                continue;
            }
            if (pl->fused2d == 10) { // N = 2^22 = 2048 x 2048, inverse
                e = launch_fused2d_inv_2k2k(p.twdl_width, src, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16r, pl->sub_col_i->h_tw.data(), pl->d_tw2d_tiles, nf, st);
                continue;
            }
            if (pl->fused2d == 11) { // ... and the pair: X in natural order in the second layout buffer between the two directions
                uint32_t *b1 = reinterpret_cast<uint32_t *>(static_cast<char *>(buf1) + off);
                e = launch_fused2d_2k2k(p.twdl_width, src, b1, b0, pl->d_tw16r, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf, st);
                if (e == hipSuccess)
                    e = launch_fused2d_inv_2k2k(p.twdl_width, b1, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16r, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf, st);
                continue;
            }
            if (pl->fused2d == 8) { // N = 2^22 = 2048 x 2048
                e = launch_fused2d_2k2k(p.twdl_width, src, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16r, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf, st);
                continue;
            }
            if (pl->fused2d == 9) { // the inverse beyond N = 2^21: X -> rows [r][k2] (k1 = brev10(r)), the N2-point inverse sub-plan, conj multiplier + column cores
                char *b1 = static_cast<char *>(buf1) + off;
                for (int j = 0; j < L; ++j) perm[j < l1 ? l2 + (l1 - 1 - j) : j - l1] = order_mem_bit(p.in_order, L, j);
                e = launch_bitperm(L, pl->in_cb, perm, src, b0, nf, st);
                if (e != hipSuccess) break;
                if ((rc = exec_core(pl->sub_row_i, b0, b1, nf << l1, st, subws)) != INTFFT_OK) break;
                e = launch_fused2d_inv_cols(l2, p.twdl_width, reinterpret_cast<const uint32_t *>(b1), reinterpret_cast<uint32_t *>(dst), pl->d_tw16f,
                                            pl->sub_col_i->h_tw.data(), pl->d_tw2d_tiles, nf, p.out_order == INTFFT_ORDER_HALVES, st);
                continue;
            }
            if (pl->fused2d == 7) { // the pair at N = 2^21: X in natural order in the second layout buffer between the two directions
                uint32_t *b1 = reinterpret_cast<uint32_t *>(static_cast<char *>(buf1) + off);
                e = launch_fused2d_cols(l2, p.twdl_width, src, b0, pl->d_tw16f, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf, p.in_order == INTFFT_ORDER_HALVES, st);
                if (e == hipSuccess) e = launch_fused2d_rows2k(p.twdl_width, b0, b1, pl->d_tw16r, pl->sub_row_f->h_tw.data(), nf, st);
                if (e == hipSuccess)
                    e = launch_fused2d_inv21(p.twdl_width, b1, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16f, pl->sub_col_f->h_tw.data(), pl->d_tw16r,
                                             pl->sub_row_f->h_tw.data(), pl->d_tw2d_tiles, nf, p.out_order == INTFFT_ORDER_HALVES, st);
                continue;
            }
            if (pl->fused2d == 6) {
                e = launch_fused2d_inv21(p.twdl_width, src, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16f, pl->sub_col_i->h_tw.data(), pl->d_tw16r,
                                         pl->sub_row_i->h_tw.data(), pl->d_tw2d_tiles, nf, p.out_order == INTFFT_ORDER_HALVES, st);
                continue;
            }
            if (pl->fused2d == 4) {
                e = launch_fused2d_inv(p.twdl_width, src, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16f, pl->sub_row_i->h_tw.data(), pl->d_tw2d_tiles, nf,
                                       p.out_order == INTFFT_ORDER_HALVES, st);
                continue;
            }
            if (pl->fused2d == 2) {
                e = launch_fused2d(p.twdl_width, src, reinterpret_cast<uint32_t *>(dst), b0, pl->d_tw16f, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf,
                                   p.in_order == INTFFT_ORDER_HALVES, st);
                continue;
            }
            e = launch_fused2d_cols(l2, p.twdl_width, src, b0, pl->d_tw16f, pl->sub_col_f->h_tw.data(), pl->d_tw2d_tiles, nf, p.in_order == INTFFT_ORDER_HALVES, st);
            if (e != hipSuccess) break;
            if (pl->d_tw16r) { // N2 = 2048, natural order out: the row cores write X themselves (one layout buffer: n2d_bufs == 1)
                e = launch_fused2d_rows2k(p.twdl_width, b0, reinterpret_cast<uint32_t *>(dst), pl->d_tw16r, pl->sub_row_f->h_tw.data(), nf, st);
                continue;
            }
            char *b1 = static_cast<char *>(buf1) + off;
            if ((rc = exec_core(pl->sub_row_f, b0, b1, nf << l1, st, subws)) != INTFFT_OK) break;
            // logical k = k1 + N1 k2 sits at [rho = brev(k1)][k2]: k bit j < l1 at in bit l2 + (l1 - 1 - j), else at j - l1
            for (int j = 0; j < L; ++j) perm[order_mem_bit(p.out_order, L, j)] = j < l1 ? l2 + (l1 - 1 - j) : j - l1;
            e = launch_bitperm(L, pl->sub_row_f->out_cb, perm, b1, dst, nf, st);
        }
        if (rc != INTFFT_OK) return rc;
        return (int)e;
    }
    for (size_t f = 0; f < batch && e == hipSuccess && rc == INTFFT_OK; f += bufs_frames) {
        const size_t nf = std::min(bufs_frames, batch - f);
        const void *src = static_cast<const char *>(d_in) + f * in_frame;
        void *dst = static_cast<char *>(d_out) + f * out_frame;
        void *cur = buf0, *oth = buf1;
        int cb = pl->in_cb;
        if (p.direction != INTFFT_INV) {
            // user (time side, logical n = n1 N2 + n2) -> [n2][n1]: n2 bit j -> bit l1 + j, n1 bit j -> bit j
            for (int j = 0; j < L; ++j) perm[j < l2 ? l1 + j : j - l2] = order_mem_bit(p.in_order, L, j);
            e = launch_bitperm(L, cb, perm, src, cur, nf, stream);
            if (e != hipSuccess) break;
            if ((rc = exec_core(pl->sub_col_f, cur, oth, nf << l2, stream, subws)) != INTFFT_OK) break;
            std::swap(cur, oth);
            cb = pl->sub_col_f->out_cb;
            // [n2][k1] -> [k1][n2]: out bit b < l2 (n2) <- in bit l1 + b; out bit b >= l2 (k1) <- in bit b - l2
            for (int b = 0; b < L; ++b) perm[b] = b < l2 ? l1 + b : b - l2;
            const StageDesc &t = pl->tw_f;
            if (fuse) { // the multiplier between the cores rides on the layout change (applied in its output layout [k1][n2])
                if ((e = launch_bitperm_tw(L, cb, perm, l2, t.mw, t.sh_a, t.sh_b, t.narrow, p.twdl_width, 0, cur, oth, nf, stream)) != hipSuccess) break;
                std::swap(cur, oth);
            } else {
                if ((e = launch_bitperm(L, cb, perm, cur, oth, nf, stream)) != hipSuccess) break;
                std::swap(cur, oth);
                if ((e = launch_twmul(cur, cb, L, l2, t.mw, t.sh_a, t.sh_b, t.narrow, 0, p.twdl_width, nf, stream)) != hipSuccess) break;
            }
            if ((rc = exec_core(pl->sub_row_f, cur, oth, nf << l1, stream, subws)) != INTFFT_OK) break;
            std::swap(cur, oth);
            cb = pl->sub_row_f->out_cb;
            if (p.direction == INTFFT_FWD) {
                // [k1][k2] -> user (frequency side, logical k = k1 + N1 k2): k bit j < l1 sits at in bit l2 + j, else at j - l1
                for (int j = 0; j < L; ++j) perm[order_mem_bit(p.out_order, L, j)] = j < l1 ? l2 + j : j - l1;
                e = launch_bitperm(L, cb, perm, cur, dst, nf, stream);
                continue;
            }
        } else {
            // user (frequency side, logical k) -> [k1][k2]
            for (int j = 0; j < L; ++j) perm[j < l1 ? l2 + j : j - l1] = order_mem_bit(p.in_order, L, j);
            if ((e = launch_bitperm(L, cb, perm, src, cur, nf, stream)) != hipSuccess) break;
        }
        if ((rc = exec_core(pl->sub_row_i, cur, oth, nf << l1, stream, subws)) != INTFFT_OK) break;
        std::swap(cur, oth);
        cb = pl->sub_row_i->out_cb;
        const StageDesc &t = pl->tw_i;
        // [k1][n2] -> [n2][k1]: out bit b < l1 (k1) <- in bit l2 + b; out bit b >= l1 (n2) <- in bit b - l1
        for (int b = 0; b < L; ++b) perm[b] = b < l1 ? l2 + b : b - l1;
        if (fuse) { // conj multiply applied in the input layout [k1][n2]
            if ((e = launch_bitperm_tw(L, cb, perm, l2, t.mw, t.sh_a, t.sh_b, t.narrow, p.twdl_width, 1, cur, oth, nf, stream)) != hipSuccess) break;
        } else {
            if ((e = launch_twmul(cur, cb, L, l2, t.mw, t.sh_a, t.sh_b, t.narrow, 1, p.twdl_width, nf, stream)) != hipSuccess) break;
            if ((e = launch_bitperm(L, cb, perm, cur, oth, nf, stream)) != hipSuccess) break;
        }
        std::swap(cur, oth);
        if ((rc = exec_core(pl->sub_col_i, cur, oth, nf << l2, stream, subws)) != INTFFT_OK) break;
        std::swap(cur, oth);
        cb = pl->sub_col_i->out_cb;
        // [n2][n1] -> user (time side, logical n = n1 N2 + n2): n bit j < l2 sits at in bit l1 + j, else at j - l2
        for (int j = 0; j < L; ++j) perm[order_mem_bit(p.out_order, L, j)] = j < l2 ? l1 + j : j - l2;
        e = launch_bitperm(L, cb, perm, cur, dst, nf, stream);
    }
    if (rc != INTFFT_OK) return rc;
    return (int)e;
}

static int exec_check(const intfft_plan *plan, const void *d_in, const void *d_out, size_t batch)
{
    if (!plan || (batch && (!d_in || !d_out))) return INTFFT_ERR_NULL;
    {   // in place only as d_in == d_out with equal containers; any other overlap of the two byte ranges would let a
        // block overwrite frames another block has not read yet
        const size_t n2 = (size_t)2 << plan->L;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(d_in), a1 = a0 + batch * n2 * (size_t)plan->in_cb;
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(d_out), b1 = b0 + batch * n2 * (size_t)plan->out_cb;
        if (a0 < b1 && b0 < a1 && (a0 != b0 || plan->in_cb != plan->out_cb)) return INTFFT_ERR_INVALID;
    }
    return INTFFT_OK;
}

int intfft_exec(intfft_plan *plan, const void *d_in, void *d_out, size_t batch, void *hip_stream)
{
    const int rc = exec_check(plan, d_in, d_out, batch);
    if (rc != INTFFT_OK || batch == 0) return rc;
    if (!plan->owns_scratch) return INTFFT_ERR_INVALID; // intfft_plan_release_scratch: intfft_exec_ws only
    DeviceGuard guard(plan->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    return exec_core(plan, d_in, d_out, batch, reinterpret_cast<hipStream_t>(hip_stream), nullptr);
}

int intfft_plan_workspace_bytes(const intfft_plan *plan, size_t batch, size_t *bytes)
{
    if (!plan || !bytes) return INTFFT_ERR_NULL;
    *bytes = ws_need(plan, batch);
    return INTFFT_OK;
}

int intfft_exec_ws(intfft_plan *plan, const void *d_in, void *d_out, size_t batch, void *d_workspace, size_t ws_bytes, void *hip_stream)
{
    int rc = exec_check(plan, d_in, d_out, batch);
    if (rc != INTFFT_OK || batch == 0) return rc;
    const size_t need = ws_need(plan, batch);
    if (need && !d_workspace) return INTFFT_ERR_NULL;
    if (need && (reinterpret_cast<uintptr_t>(d_workspace) & 255)) return INTFFT_ERR_INVALID;
    {   // the workspace must not overlap the user arrays
        const size_t n2 = (size_t)2 << plan->L;
        const uintptr_t w0 = reinterpret_cast<uintptr_t>(d_workspace), w1 = w0 + ws_bytes;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(d_in), a1 = a0 + batch * n2 * (size_t)plan->in_cb;
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(d_out), b1 = b0 + batch * n2 * (size_t)plan->out_cb;
        if (need && ((w0 < a1 && a0 < w1) || (w0 < b1 && b0 < w1))) return INTFFT_ERR_INVALID;
    }
    DeviceGuard guard(plan->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    char *ws = static_cast<char *>(d_workspace);
    if (need <= ws_bytes) return exec_core(plan, d_in, d_out, batch, stream, need ? ws : nullptr);
    // a smaller workspace: the largest sub-batch it serves (ws_need is monotone), sub-batch after sub-batch on the caller's stream
    if (ws_need(plan, 1) > ws_bytes) return INTFFT_ERR_INVALID;
    size_t lo = 1, hi = batch; // ws_need(lo) <= ws_bytes < ws_need(hi)
    while (hi - lo > 1) {
        const size_t mid = lo + (hi - lo) / 2;
        if (ws_need(plan, mid) <= ws_bytes) lo = mid;
        else hi = mid;
    }
    const size_t in_frame = ((size_t)2 << plan->L) * (size_t)plan->in_cb, out_frame = ((size_t)2 << plan->L) * (size_t)plan->out_cb;
    for (size_t f = 0; f < batch && rc == INTFFT_OK; f += lo)
        rc = exec_core(plan, static_cast<const char *>(d_in) + f * in_frame, static_cast<char *>(d_out) + f * out_frame, std::min(lo, batch - f), stream, ws);
    return rc;
}

int intfft_plan_release_scratch(intfft_plan *plan)
{
    if (!plan) return INTFFT_ERR_NULL;
    DeviceGuard guard(plan->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    const hipError_t e = hipDeviceSynchronize(); // nothing of an earlier intfft_exec may still be using the buffers
    for (void **b : {&plan->d_scratch, &plan->d_scratch2, &plan->buf2d[0], &plan->buf2d[1], &plan->pair_buf})
        if (*b) {
            (void)hipFree(*b);
            *b = nullptr;
        }
    plan->owns_scratch = false;
    for (intfft_plan *sp : {plan->sub_col_f, plan->sub_row_f, plan->sub_row_i, plan->sub_col_i, plan->pair_f, plan->pair_i})
        if (sp) (void)intfft_plan_release_scratch(sp);
    return (int)e;
}

// One call on the plan's own scratch (ws == nullptr) or on a caller-supplied workspace of ws_need(plan, batch) bytes.  Reads the plan
// only (execution contexts come from its pool), so calls with distinct workspaces may run concurrently.
static int exec_core(intfft_plan *plan, const void *d_in, void *d_out, size_t batch, hipStream_t stream, char *ws)
{
    if (plan->is2d) return exec_2d(plan, d_in, d_out, batch, stream, ws);
    if (plan->lanes_mode == 3) { // USE_FLY = 0: wrap to DATA_WIDTH in the output container, then in_order -> out_order
        const int L = plan->L, zext = plan->p.format ? 1 : 0;
        const size_t in_frame = ((size_t)2 << L) * (size_t)plan->in_cb, out_frame = ((size_t)2 << L) * (size_t)plan->out_cb;
        if (!plan->pair_frames)
            return (int)launch_convert(plan->in_cb, plan->out_cb, plan->p.data_width, zext, d_in, d_out, batch * ((size_t)2 << L), stream);
        const size_t pf = ws ? std::min(plan->pair_frames, batch) : plan->pair_frames;
        void *const mid = ws ? ws : plan->pair_buf;
        const int *const perm = plan->bypass_perm;
        for (size_t f = 0; f < batch; f += pf) {
            const size_t nf = std::min(pf, batch - f);
            int rc = (int)launch_convert(plan->in_cb, plan->out_cb, plan->p.data_width, zext, static_cast<const char *>(d_in) + f * in_frame, mid,
                                         nf * ((size_t)2 << L), stream);
            if (rc == INTFFT_OK) rc = (int)launch_bitperm(L, plan->out_cb, perm, mid, static_cast<char *>(d_out) + f * out_frame, nf, stream);
            if (rc != INTFFT_OK) return rc;
        }
        return INTFFT_OK;
    }
    if (plan->lanes_mode) { // BITREV_LANES composite: the BITREV twin and one bit permutation through the middle buffer, chunk by chunk
        const int L = plan->L;
        const size_t in_frame = ((size_t)2 << L) * (size_t)plan->in_cb, out_frame = ((size_t)2 << L) * (size_t)plan->out_cb;
        const size_t pf = ws ? std::min(plan->pair_frames, batch) : plan->pair_frames;
        void *const mid = ws ? ws : plan->pair_buf;
        char *const subws = ws ? ws + ws_align(pf * plan->scratch_frame_bytes) : nullptr;
        // memory index m_B (BITREV) and m_S (BITREV_LANES) of one sample: m_S = (m_B & 1) << (L-1) | m_B >> 1
        int perm[24]; // m_in bit perm[b] = m_out bit b
        for (int b = 0; b < L; ++b) perm[b] = plan->lanes_mode == 1 ? (b == L - 1 ? 0 : b + 1) : (b == 0 ? L - 1 : b - 1);
        for (size_t f = 0; f < batch; f += pf) {
            const size_t nf = std::min(pf, batch - f);
            const char *src = static_cast<const char *>(d_in) + f * in_frame;
            char *dst = static_cast<char *>(d_out) + f * out_frame;
            int rc;
            if (plan->lanes_mode == 1) {
                rc = exec_core(plan->pair_f, src, mid, nf, stream, subws);
                if (rc == INTFFT_OK) rc = (int)launch_bitperm(L, plan->out_cb, perm, mid, dst, nf, stream);
            } else {
                rc = (int)launch_bitperm(L, plan->in_cb, perm, src, mid, nf, stream);
                if (rc == INTFFT_OK) rc = exec_core(plan->pair_f, mid, dst, nf, stream, subws);
            }
            if (rc != INTFFT_OK) return rc;
        }
        return INTFFT_OK;
    }
    if (plan->is_pair) { // composite pair: forward sub-plan -> middle buffer -> inverse sub-plan, chunk by chunk
        const size_t in_frame = ((size_t)2 << plan->L) * (size_t)plan->in_cb, out_frame = ((size_t)2 << plan->L) * (size_t)plan->out_cb;
        const size_t pf = ws ? std::min(plan->pair_frames, batch) : plan->pair_frames;
        void *const mid = ws ? ws : plan->pair_buf;
        char *const subws = ws ? ws + ws_align(pf * plan->scratch_frame_bytes) : nullptr;
        for (size_t f = 0; f < batch; f += pf) {
            const size_t nf = std::min(pf, batch - f);
            int rc = exec_core(plan->pair_f, static_cast<const char *>(d_in) + f * in_frame, mid, nf, stream, subws);
            if (rc == INTFFT_OK) rc = exec_core(plan->pair_i, mid, static_cast<char *>(d_out) + f * out_frame, nf, stream, subws);
            if (rc != INTFFT_OK) return rc;
        }
        return INTFFT_OK;
    }
    if (plan->fastsmall)
        return (int)launch_fastsmall(plan->p.log2n, plan->p.direction, plan->p.rndmode, plan->p.twdl_width, d_in, d_out,
                                     plan->h_tw.data(), batch, stream, plan->p.data_width);
    if (plan->w32inv)
        return (int)launch_w32inv(plan->p.log2n, plan->p.format ? 2 : plan->p.rndmode, plan->w32args, d_in, d_out, plan->d_tw,
                                  plan->h_tw.data(), batch, stream,
                                  (plan->p.out_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.in_order == INTFFT_ORDER_BITREV ? 2 : 0));
    if (plan->fast4096w)
        return (int)launch_fast4096w(plan->p.log2n, plan->p.format ? 2 : plan->p.rndmode, plan->w32args, d_in, d_out, plan->d_tw,
                                     plan->h_tw.data(), batch, stream,
                                     (plan->p.in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.out_order == INTFFT_ORDER_BITREV ? 2 : 0));
    if (plan->fastw64b)
        return (int)launch_fastw64b(plan->p.log2n, plan->p.direction, plan->p.format ? RND_UNSCALED : plan->p.rndmode ? RND_ROUND : RND_TRUNC, plan->st64, plan->in_cb,
                                    plan->p.data_width, d_in, d_out, plan->d_tw, plan->h_tw.data(), batch, stream,
                                    plan->p.direction == INTFFT_INV ? ((plan->p.out_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.in_order == INTFFT_ORDER_BITREV ? 2 : 0))
                                                                    : ((plan->p.in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.out_order == INTFFT_ORDER_BITREV ? 2 : 0)));
    if (plan->fastw64)
        return (int)launch_fastw64(plan->p.log2n, plan->p.direction, plan->p.format ? RND_UNSCALED : plan->p.rndmode ? RND_ROUND : RND_TRUNC, plan->st64, plan->in_cb, plan->p.data_width, d_in,
                                   d_out, plan->d_tw, plan->h_tw.data(), batch, stream,
                                   plan->p.direction == INTFFT_INV ? ((plan->p.out_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.in_order == INTFFT_ORDER_BITREV ? 2 : 0))
                                                                   : ((plan->p.in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.out_order == INTFFT_ORDER_BITREV ? 2 : 0)));
    if (plan->fastw32)
        return (int)launch_fastw32(plan->p.log2n, plan->p.format ? 2 : plan->p.rndmode, plan->w32args, d_in, d_out, plan->d_tw,
                                   plan->h_tw.data(), batch, stream,
                                   (plan->p.in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.out_order == INTFFT_ORDER_BITREV ? 2 : 0));
    if (plan->fast1024ux)
        return (int)launch_fast1024ux(plan->p.log2n, plan->p.direction, plan->p.twdl_width, plan->uxargs, d_in, d_out,
                                      plan->d_tw, plan->h_tw.data(), batch, stream,
                                      (plan->p.out_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.in_order == INTFFT_ORDER_BITREV ? 2 : 0));
    if (plan->fast1024u)
        return (int)launch_fast1024u(plan->p.log2n, plan->p.twdl_width, d_in, d_out, plan->d_tw, plan->h_tw.data(), batch, stream,
                                     (plan->p.in_order == INTFFT_ORDER_HALVES ? 1 : 0) | (plan->p.out_order == INTFFT_ORDER_BITREV ? 2 : 0));
    if (plan->fast1024)
        return (int)launch_fast1024(plan->fargs, d_in, d_out, plan->d_tw, plan->h_tw.data(), batch, stream);
    if (plan->fast1024x)
        return (int)launch_fast1024x(plan->p.log2n, plan->p.direction, plan->p.twdl_width,
                                     plan->p.in_order == INTFFT_ORDER_BITREV ? 1 : plan->p.in_order == INTFFT_ORDER_BITREV_LANES ? 2 : 0,
                                     plan->p.out_order == INTFFT_ORDER_HALVES, d_in, d_out, plan->d_tw,
                                     plan->h_tw.data(), batch, stream, plan->p.rndmode, plan->p.data_width);
    if (plan->fast16k)
        return (int)launch_fast16k(plan->p.log2n, plan->p.direction, plan->p.twdl_width, d_in, d_out, plan->d_tw16f, plan->h_tw.data(), batch, stream,
                                   plan->p.data_width, plan->p.rndmode,
                                   ((plan->p.direction == INTFFT_FWD ? plan->p.in_order : plan->p.out_order) == INTFFT_ORDER_HALVES ? 1 : 0) |
                                       ((plan->p.direction == INTFFT_FWD ? plan->p.out_order : plan->p.in_order) == INTFFT_ORDER_BITREV ? 2 : 0) |
                                       ((plan->p.direction == INTFFT_FWD ? plan->p.out_order : plan->p.in_order) == INTFFT_ORDER_BITREV_LANES ? 4 : 0));
    if (plan->fast4096)
        return (int)launch_fast4096(plan->p.log2n, plan->p.direction, plan->p.twdl_width,
                                    [&] { // the frequency-side order: 1 BITREV, 2 BITREV_LANES
                                        const int o = plan->p.direction == INTFFT_FWD ? plan->p.out_order : plan->p.in_order;
                                        return o == INTFFT_ORDER_BITREV ? 1 : o == INTFFT_ORDER_BITREV_LANES ? 2 : 0;
                                    }(),
                                    plan->p.direction == INTFFT_FWD ? plan->p.in_order == INTFFT_ORDER_HALVES
                                                                    : plan->p.out_order == INTFFT_ORDER_HALVES,
                                    d_in, d_out, plan->d_tw, plan->h_tw.data(), batch, stream, plan->p.rndmode, plan->p.data_width);

    const size_t N = (size_t)1 << plan->L;
    const size_t in_frame = N * 2 * (size_t)plan->in_cb, out_frame = N * 2 * (size_t)plan->out_cb;
    const size_t np = plan->passes.size();
    const size_t chunk = (np > 1 || plan->big20 || plan->bigw || plan->wide16) ? plan->scratch_frames : batch;
    // more than one chunk: odd chunks run on a pooled side stream with the second scratch half (fork here, join on every exit path)
    hipStream_t const user_stream = stream;
    void *const scratch0 = ws ? ws : plan->d_scratch;
    void *const scratch1 = ws ? (plan->dual_scratch ? ws + ws_align(plan->scratch_frames * plan->scratch_frame_bytes) : nullptr) : plan->d_scratch2;
    SideStream side(plan, user_stream);
    if (plan->dual_scratch && scratch1 && batch > chunk) {
        const hipError_t e = side.fork();
        if (e != hipSuccess) return (int)e;
    }
    const bool dual = side.on;
    size_t ci = 0;
    for (size_t f = 0; f < batch; f += chunk, ++ci) {
        const size_t nf = std::min(chunk, batch - f);
        const void *src = static_cast<const char *>(d_in) + f * in_frame;
        void *dst = static_cast<char *>(d_out) + f * out_frame;
        const bool odd = dual && (ci & 1);
        stream = odd ? side.c.side : user_stream;
        void *const scratch = odd ? scratch1 : scratch0;
        // (also for one-frame batches: a lone N = 8192 frame takes 10 us here, 21-53 us as one workgroup of the generic pass)
        if (plan->bigw) {
            const hipError_t e = launch_bigw(plan->p.log2n, plan->p.format ? 2 : plan->p.rndmode, plan->w32args, src, dst,
                                             scratch, plan->d_tw, plan->h_tw.data(), nf, stream);
            if (e != hipSuccess) return (int)e;
            continue;
        }
        if (plan->widelong) {
            const hipError_t e = launch_widelong(plan->p.log2n, plan->wargs, plan->in_cb, src, dst, scratch, plan->d_tw, plan->h_tw.data(), nf, stream, plan->p.direction);
            if (e != hipSuccess) return (int)e;
            continue;
        }
        if (plan->wide16) {
            const hipError_t e = launch_wide16(plan->p.log2n, plan->wargs, src, dst, scratch, plan->d_tw, plan->h_tw.data(), nf,
                                               stream, plan->p.direction);
            if (e != hipSuccess) return (int)e;
            continue;
        }
        if (plan->big20) {
            const hipError_t e = plan->p.direction == INTFFT_INV
                                     ? launch_biginv(plan->p.log2n, plan->p.twdl_width, plan->p.in_order == INTFFT_ORDER_BITREV,
                                                     plan->p.out_order == INTFFT_ORDER_HALVES, plan->big_two_pass, src, dst, scratch, plan->d_tw,
                                                     plan->d_tw16f, plan->h_tw.data(), nf, stream, plan->p.data_width, plan->p.rndmode)
                                 : plan->p.direction == INTFFT_PAIR
                                     ? launch_bigpair(plan->p.log2n, plan->p.twdl_width, plan->big_pair256, src, dst, scratch, plan->d_tw,
                                                      plan->d_tw16f, plan->h_tw.data(), nf, stream, plan->p.data_width, plan->p.rndmode)
                                     : launch_big20(plan->p.log2n, plan->p.twdl_width, plan->p.in_order == INTFFT_ORDER_HALVES,
                                                    plan->p.out_order == INTFFT_ORDER_BITREV, plan->big_two_pass, src, dst, scratch, plan->d_tw,
                                                    plan->d_tw16f, plan->h_tw.data(), nf, stream, plan->p.data_width, plan->p.rndmode);
            if (e != hipSuccess) return (int)e;
            continue;
        }
        for (size_t i = 0; i < np; ++i) {
            const PassArgs &a = plan->passes[i];
            const void *pin = a.in_mode == IO_USER ? src : scratch;
            void *pout = a.out_mode == IO_USER ? dst : scratch;
            const hipError_t e = plan->word == 2
                                     ? launch_pass16(a, pin, pout, plan->d_tw16f, plan->d_tw16i, nf, plan->p.twdl_width, stream)
                                     : launch_pass(a, a.word, pin, pout, plan->d_tw, nf, stream, plan->d_tw2d);
            if (e != hipSuccess) return (int)e;
        }
    }
    return INTFFT_OK;
}

static void free_stream_state(intfft_plan *pl)
{
    for (int i = 0; i < 2; ++i) {
        if (pl->slot_in[i]) (void)hipFree(pl->slot_in[i]);
        if (pl->slot_out[i]) (void)hipFree(pl->slot_out[i]);
        if (pl->ev_up[i]) (void)hipEventDestroy(pl->ev_up[i]);
        if (pl->ev_comp[i]) (void)hipEventDestroy(pl->ev_comp[i]);
        if (pl->ev_down[i]) (void)hipEventDestroy(pl->ev_down[i]);
        pl->slot_in[i] = pl->slot_out[i] = nullptr;
        pl->ev_up[i] = pl->ev_comp[i] = pl->ev_down[i] = nullptr;
    }
    if (pl->s_up) (void)hipStreamDestroy(pl->s_up);
    if (pl->s_comp) (void)hipStreamDestroy(pl->s_comp);
    if (pl->s_down) (void)hipStreamDestroy(pl->s_down);
    pl->s_up = pl->s_comp = pl->s_down = nullptr;
    pl->slot_frames = 0;
}

int intfft_exec_host(intfft_plan *plan, const void *h_in, void *h_out, size_t batch, size_t chunk_frames)
{
    if (!plan || (batch && (!h_in || !h_out))) return INTFFT_ERR_NULL;
    if (batch == 0) return INTFFT_OK;
    if (!plan->owns_scratch) return INTFFT_ERR_INVALID; // intfft_plan_release_scratch: intfft_exec_ws only (before any stream / slot is created)
    DeviceGuard guard(plan->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    const size_t N = (size_t)1 << plan->L;
    const size_t in_frame = N * 2 * (size_t)plan->in_cb, out_frame = N * 2 * (size_t)plan->out_cb;
    int lib_rc = INTFFT_OK; // a negative library status of the inner intfft_exec goes back to the caller unchanged
    if (chunk_frames == 0) chunk_frames = std::max<size_t>(1, ((size_t)64 << 20) / std::max(in_frame, out_frame));
    chunk_frames = std::min(chunk_frames, batch);
    hipError_t e = hipSuccess;
#define INTFFT_TRY(x) do { e = (x); if (e != hipSuccess) goto fail; } while (0)
    if (plan->slot_frames < chunk_frames) {
        free_stream_state(plan);
        INTFFT_TRY(hipStreamCreateWithFlags(&plan->s_up, hipStreamNonBlocking));
        INTFFT_TRY(hipStreamCreateWithFlags(&plan->s_comp, hipStreamNonBlocking));
        INTFFT_TRY(hipStreamCreateWithFlags(&plan->s_down, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            INTFFT_TRY(hipMalloc(&plan->slot_in[i], chunk_frames * in_frame));
            INTFFT_TRY(hipMalloc(&plan->slot_out[i], chunk_frames * out_frame));
            INTFFT_TRY(hipEventCreateWithFlags(&plan->ev_up[i], hipEventDisableTiming));
            INTFFT_TRY(hipEventCreateWithFlags(&plan->ev_comp[i], hipEventDisableTiming));
            INTFFT_TRY(hipEventCreateWithFlags(&plan->ev_down[i], hipEventDisableTiming));
        }
        plan->slot_frames = chunk_frames;
    }
    {
        // The caller's buffers are used as they are.  Pinned buffers (hipHostMalloc, or registered by their owner) make
        // the copies truly asynchronous; pageable buffers still work (the runtime stages them, with less overlap).
        // This function does NOT hipHostRegister the caller's memory itself: registering malloc-heap memory for the
        // duration of a call left stale GPU mappings behind on this stack once the allocator recycled those pages, and
        // later unrelated pageable copies faulted ("Memory access fault by GPU node", found by soaking the test suite).
        size_t i = 0;
        for (size_t f = 0; f < batch; f += chunk_frames, ++i) {
            const int slot = (int)(i & 1);
            const size_t nf = std::min(chunk_frames, batch - f);
            // upload: the slot's input buffer is free once the transform that last read it is done
            if (i >= 2) e = hipStreamWaitEvent(plan->s_up, plan->ev_comp[slot], 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(plan->slot_in[slot], static_cast<const char *>(h_in) + f * in_frame, nf * in_frame,
                                   hipMemcpyHostToDevice, plan->s_up);
            if (e == hipSuccess) e = hipEventRecord(plan->ev_up[slot], plan->s_up);
            // transform: needs the upload, and the previous download out of this slot's output buffer
            if (e == hipSuccess) e = hipStreamWaitEvent(plan->s_comp, plan->ev_up[slot], 0);
            if (e == hipSuccess && i >= 2) e = hipStreamWaitEvent(plan->s_comp, plan->ev_down[slot], 0);
            if (e == hipSuccess) {
                const int rc = intfft_exec(plan, plan->slot_in[slot], plan->slot_out[slot], nf, plan->s_comp);
                if (rc > 0) e = (hipError_t)rc;
                else if (rc < 0) lib_rc = rc, e = hipErrorUnknown;
            }
            if (e == hipSuccess) e = hipEventRecord(plan->ev_comp[slot], plan->s_comp);
            // download
            if (e == hipSuccess) e = hipStreamWaitEvent(plan->s_down, plan->ev_comp[slot], 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(static_cast<char *>(h_out) + f * out_frame, plan->slot_out[slot], nf * out_frame,
                                   hipMemcpyDeviceToHost, plan->s_down);
            if (e == hipSuccess) e = hipEventRecord(plan->ev_down[slot], plan->s_down);
            if (e != hipSuccess) break;
        }
        const hipError_t e2 = hipStreamSynchronize(plan->s_down);
        const hipError_t e3 = hipStreamSynchronize(plan->s_comp);
        const hipError_t e4 = hipStreamSynchronize(plan->s_up);
        if (e == hipSuccess) e = e2 != hipSuccess ? e2 : e3 != hipSuccess ? e3 : e4;
    }
    if (lib_rc != INTFFT_OK) return lib_rc;
    if (e == hipSuccess) return INTFFT_OK;
fail:
#undef INTFFT_TRY
    return (int)e;
}


// staging buffers + stream of one shard plan (grow only); peer access root <-> pl->device
static hipError_t shard_state(intfft_plan *pl, int root_device, size_t in_bytes, size_t out_bytes)
{
    hipError_t e = hipSuccess;
    if (!pl->s_shard) e = hipStreamCreateWithFlags(&pl->s_shard, hipStreamNonBlocking);
    if (e == hipSuccess && !pl->s_shard2) e = hipStreamCreateWithFlags(&pl->s_shard2, hipStreamNonBlocking);
    if (e == hipSuccess && !pl->s_call) e = hipStreamCreateWithFlags(&pl->s_call, hipStreamNonBlocking);
    for (int k = 0; k < SHARD_EVENTS && e == hipSuccess; ++k)
        if (!pl->shard_ev[k]) e = hipEventCreateWithFlags(&pl->shard_ev[k], hipEventDisableTiming);
    if (e != hipSuccess) return e;
    if (pl->device != root_device && pl->shard_peer != root_device) {
        int can = 0;
        if ((e = hipDeviceCanAccessPeer(&can, pl->device, root_device)) != hipSuccess) return e;
        if (can) { // direct xGMI copies; without it hipMemcpyPeerAsync still works (staged by the runtime)
            e = hipDeviceEnablePeerAccess(root_device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(), e = hipSuccess;
            if (e != hipSuccess) return e;
        }
        pl->shard_peer = root_device;
    }
    // (growing a staging buffer frees the old one: hipFree waits for the device, so an earlier asynchronous call that still uses it has
    // finished by then)
    if (pl->shard_in_bytes < in_bytes) {
        if (pl->shard_in) (void)hipFree(pl->shard_in);
        pl->shard_in = nullptr, pl->shard_in_bytes = 0;
        if ((e = hipMalloc(&pl->shard_in, in_bytes)) != hipSuccess) return e;
        pl->shard_in_bytes = in_bytes;
    }
    if (pl->shard_out_bytes < out_bytes) {
        if (pl->shard_out) (void)hipFree(pl->shard_out);
        pl->shard_out = nullptr, pl->shard_out_bytes = 0;
        if ((e = hipMalloc(&pl->shard_out, out_bytes)) != hipSuccess) return e;
        pl->shard_out_bytes = out_bytes;
    }
    return hipSuccess;
}

static int shard_check(intfft_plan *const *plans, int nplans, int root)
{
    if (!plans || nplans <= 0 || root < 0 || root >= nplans) return INTFFT_ERR_INVALID;
    for (int i = 0; i < nplans; ++i) {
        if (!plans[i]) return INTFFT_ERR_NULL;
        if (std::memcmp(&plans[i]->p, &plans[0]->p, sizeof(intfft_params)) != 0) return INTFFT_ERR_INVALID;
        for (int j = 0; j < i; ++j)
            if (plans[j] == plans[i]) return INTFFT_ERR_INVALID; // a plan owns its scratch: one shard at a time
    }
    return INTFFT_OK;
}

static size_t shard_frames(size_t batch, int nplans, int i)
{
    const size_t base = batch / (size_t)nplans, rem = batch % (size_t)nplans;
    return base + ((size_t)i >= (size_t)nplans - rem ? 1 : 0);
}

int intfft_shard_prepare(intfft_plan *const *plans, int nplans, int root, size_t max_batch)
{
    const int rc = shard_check(plans, nplans, root);
    if (rc != INTFFT_OK) return rc;
    intfft_plan *rp = plans[root];
    const size_t N = (size_t)1 << rp->L;
    const size_t in_frame = N * 2 * (size_t)rp->in_cb, out_frame = N * 2 * (size_t)rp->out_cb;
    for (int i = 0; i < nplans; ++i) {
        intfft_plan *pl = plans[i];
        DeviceGuard g(pl->device);
        if (!g.ok) return INTFFT_ERR_NO_DEVICE;
        const size_t nf = i == root ? 0 : shard_frames(max_batch, nplans, i);
        const hipError_t e = shard_state(pl, rp->device, nf * in_frame, nf * out_frame);
        if (e != hipSuccess) return (int)e;
        if (i != root && pl->device != rp->device) { // and the reverse direction (the gather writes into the root's memory)
            DeviceGuard gr(rp->device);
            int can = 0;
            if (gr.ok && hipDeviceCanAccessPeer(&can, rp->device, pl->device) == hipSuccess && can)
                if (hipDeviceEnablePeerAccess(pl->device, 0) == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    return INTFFT_OK;
}

int intfft_shard_set_transport(intfft_plan *const *plans, int nplans, int root, int transport)
{
    const int rc = shard_check(plans, nplans, root);
    if (rc != INTFFT_OK) return rc;
    if (transport != INTFFT_TRANSPORT_PEER && transport != INTFFT_TRANSPORT_RCCL) return INTFFT_ERR_INVALID;
    // drop what these plans had: a plan that owns a set releases all of it (every member of that set, in this call's set or not, goes
    // back to peer copies); a member of somebody else's set leaves that set
    for (int i = 0; i < nplans; ++i) {
        if (plans[i]->rccl_owner == plans[i]) rccl_release(plans[i]);
        else rccl_leave(plans[i]);
    }
    if (transport == INTFFT_TRANSPORT_PEER) return INTFFT_OK;
    Rccl &r = rccl();
    if (!r.ok) return INTFFT_ERR_TRANSPORT;
    std::vector<int> devs(nplans);
    for (int i = 0; i < nplans; ++i) {
        devs[i] = plans[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return INTFFT_ERR_TRANSPORT; // RCCL: one rank per device
    }
    std::vector<void *> comms(nplans, nullptr);
    const int st = r.comm_init_all(comms.data(), nplans, devs.data());
    if (st != 0) {
        rccl_report("ncclCommInitAll", st);
        return INTFFT_ERR_TRANSPORT;
    }
    static std::atomic<uint64_t> next_set{1};
    const uint64_t id = next_set.fetch_add(1);
    for (int i = 0; i < nplans; ++i) {
        plans[i]->rccl_comm = comms[i], plans[i]->rccl_rank = i, plans[i]->rccl_nranks = nplans;
        plans[i]->rccl_set = id, plans[i]->rccl_owner = plans[0];
    }
    plans[0]->rccl_owned = comms;
    plans[0]->rccl_members.assign(plans, plans + nplans);
    return INTFFT_OK;
}

// How many pieces a shard is cut into so that the gather of piece k overlaps the scatter of piece k + 1 on the full-duplex links
// (end to end is link-bound; serial scatter -> transform -> gather pays both directions one after the other).
// Pieces stay >= 2 MiB of input so that a piece still fills the chip and a link transfer is not latency-bound.
static int shard_pieces(size_t frames, size_t in_frame)
{
    int want = 4;
    if (const char *e = diag_env("INTFFT_SHARD_PIECES")) want = std::max(1, std::min(SHARD_EVENTS - 3, atoi(e)));
    else want = (int)std::max<size_t>(1, std::min<size_t>(4, frames * in_frame / ((size_t)2 << 20)));
    return (int)std::min<size_t>((size_t)want, std::max<size_t>(1, frames));
}
static void piece_range(size_t frames, int pieces, int k, size_t &first, size_t &cnt)
{
    const size_t base = frames / (size_t)pieces, rem = frames % (size_t)pieces;
    first = (size_t)k * base + std::min<size_t>((size_t)k, rem);
    cnt = base + ((size_t)k < rem ? 1 : 0);
}

// The asynchronous core of intfft_exec_sharded: everything is enqueued behind an event recorded on `user` (a stream of the root
// device) and `user` waits for every stream that was used; no host synchronisation.  shard_ev: [0] entry (root plan), [1] done (per
// plan: what the next call's first operation on this plan's streams waits for), [2] its first stream done, [3 + k] transform of piece k done.
static int exec_sharded_enqueue(intfft_plan *const *plans, int nplans, int root, const void *d_in, void *d_out, size_t batch, hipStream_t user, bool use_rccl)
{
    Rccl &r = rccl();
    intfft_plan *rp = plans[root];
    const size_t N = (size_t)1 << rp->L;
    const size_t in_frame = N * 2 * (size_t)rp->in_cb, out_frame = N * 2 * (size_t)rp->out_cb;
    hipError_t e = hipSuccess;
    int rc = INTFFT_OK, nc = 0;
    const char *nc_what = "";
    auto ck = [&](int st, const char *what) { // the first failing RCCL call is the one reported
        if (st != 0 && nc == 0) nc = st, nc_what = what;
    };
    std::vector<size_t> first(nplans, 0), cnt(nplans, 0);
    size_t widest = 0;
    for (int i = 0, start = 0; i < nplans; ++i) {
        cnt[i] = shard_frames(batch, nplans, i);
        first[i] = (size_t)start;
        start += (int)cnt[i];
        if (i != root) widest = std::max(widest, cnt[i]);
    }
    const int pieces = widest ? shard_pieces(widest, in_frame) : 1; // one schedule for all peers (shard sizes differ by one frame at most)
    for (int i = 0; i < nplans && e == hipSuccess; ++i) { // staging + streams + events (no-op after intfft_shard_prepare)
        DeviceGuard g(plans[i]->device);
        if (!g.ok) return INTFFT_ERR_NO_DEVICE;
        e = shard_state(plans[i], rp->device, i == root ? 0 : cnt[i] * in_frame, i == root ? 0 : cnt[i] * out_frame);
    }
    if (e != hipSuccess) return (int)e;
    {   // entry: every stream of the call starts behind the caller's stream and behind the previous call on these plans
        DeviceGuard g(rp->device);
        if (!g.ok) return INTFFT_ERR_NO_DEVICE;
        if ((e = hipEventRecord(rp->shard_ev[0], user)) != hipSuccess) return (int)e;
    }
    for (int i = 0; i < nplans && e == hipSuccess; ++i) {
        intfft_plan *pl = plans[i];
        DeviceGuard g(pl->device);
        for (hipStream_t st : {pl->s_shard, pl->s_shard2}) {
            if (e == hipSuccess) e = hipStreamWaitEvent(st, rp->shard_ev[0], 0);
            if (e == hipSuccess && pl->shard_used) e = hipStreamWaitEvent(st, pl->shard_ev[1], 0);
        }
    }
    if (e != hipSuccess) return (int)e;
    // the root's own shard: one call on its second stream, beside its sends / the peers' copies
    if (cnt[root]) {
        DeviceGuard g(rp->device);
        rc = intfft_exec(rp, static_cast<const char *>(d_in) + first[root] * in_frame, static_cast<char *>(d_out) + first[root] * out_frame, cnt[root], rp->s_shard2);
    }
    if (!use_rccl) {
        // peer copies: per peer, piece k goes in and is transformed on its first stream, and comes back on its second one -- the
        // copy back of piece k beside the copy in of piece k + 1 (the links are full duplex)
        for (int k = 0; k < pieces && e == hipSuccess && rc == INTFFT_OK; ++k)
            for (int i = 0; i < nplans && e == hipSuccess && rc == INTFFT_OK; ++i) {
                intfft_plan *pl = plans[i];
                size_t pf, pn;
                piece_range(cnt[i], pieces, k, pf, pn);
                if (i == root || pn == 0) continue;
                DeviceGuard g(pl->device);
                if (!g.ok) {
                    rc = INTFFT_ERR_NO_DEVICE;
                    break;
                }
                const char *src = static_cast<const char *>(d_in) + (first[i] + pf) * in_frame;
                char *dst = static_cast<char *>(d_out) + (first[i] + pf) * out_frame;
                char *sin = static_cast<char *>(pl->shard_in) + pf * in_frame, *sout = static_cast<char *>(pl->shard_out) + pf * out_frame;
                e = hipMemcpyPeerAsync(sin, pl->device, src, rp->device, pn * in_frame, pl->s_shard);
                if (e != hipSuccess) break;
                rc = intfft_exec(pl, sin, sout, pn, pl->s_shard);
                if (rc != INTFFT_OK) break;
                e = hipEventRecord(pl->shard_ev[3 + k], pl->s_shard);
                if (e == hipSuccess) e = hipStreamWaitEvent(pl->s_shard2, pl->shard_ev[3 + k], 0);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(dst, rp->device, sout, pl->device, pn * out_frame, pl->s_shard2);
            }
    } else {
        // RCCL: step t = ONE group of { scatter of piece t, gather of piece t - 1 } -- the root sends to and receives from every peer in
        // the same group, so all its links run in both directions at once --, then the peers transform piece t.  (Operations of one
        // communicator execute in issue order whatever their streams, so the overlap has to be inside a group.)
        for (int t = 0; t <= pieces && nc == 0 && rc == INTFFT_OK; ++t) {
            ck(r.group_start(), "ncclGroupStart");
            for (int i = 0; i < nplans; ++i) {
                if (i == root) continue;
                size_t pf, pn;
                if (t < pieces) {
                    piece_range(cnt[i], pieces, t, pf, pn);
                    if (pn) {
                        {
                            DeviceGuard g(rp->device);
                            ck(r.send(static_cast<const char *>(d_in) + (first[i] + pf) * in_frame, pn * in_frame, NCCL_INT8, i, rp->rccl_comm, rp->s_shard), "ncclSend (scatter)");
                        }
                        DeviceGuard g(plans[i]->device);
                        ck(r.recv(static_cast<char *>(plans[i]->shard_in) + pf * in_frame, pn * in_frame, NCCL_INT8, root, plans[i]->rccl_comm, plans[i]->s_shard), "ncclRecv (scatter)");
                    }
                }
                if (t >= 1) {
                    piece_range(cnt[i], pieces, t - 1, pf, pn);
                    if (pn) {
                        {
                            DeviceGuard g(plans[i]->device);
                            ck(r.send(static_cast<const char *>(plans[i]->shard_out) + pf * out_frame, pn * out_frame, NCCL_INT8, root, plans[i]->rccl_comm, plans[i]->s_shard), "ncclSend (gather)");
                        }
                        DeviceGuard g(rp->device);
                        ck(r.recv(static_cast<char *>(d_out) + (first[i] + pf) * out_frame, pn * out_frame, NCCL_INT8, i, rp->rccl_comm, rp->s_shard), "ncclRecv (gather)");
                    }
                }
            }
            ck(r.group_end(), "ncclGroupEnd");
            for (int i = 0; i < nplans && t < pieces && nc == 0 && rc == INTFFT_OK; ++i) { // stream order: behind the receive of piece t
                size_t pf, pn;
                piece_range(cnt[i], pieces, t, pf, pn);
                if (i == root || pn == 0) continue;
                DeviceGuard g(plans[i]->device);
                rc = intfft_exec(plans[i], static_cast<char *>(plans[i]->shard_in) + pf * in_frame, static_cast<char *>(plans[i]->shard_out) + pf * out_frame, pn, plans[i]->s_shard);
            }
        }
    }
    // exit: the caller's stream waits for every stream of the call, also after an error (what was enqueued still runs)
    for (int i = 0; i < nplans; ++i) {
        intfft_plan *pl = plans[i];
        DeviceGuard g(pl->device);
        hipError_t ee = hipEventRecord(pl->shard_ev[2], pl->s_shard);
        if (ee == hipSuccess) ee = hipStreamWaitEvent(pl->s_shard2, pl->shard_ev[2], 0);
        if (ee == hipSuccess) ee = hipEventRecord(pl->shard_ev[1], pl->s_shard2); // done = both streams of this plan
        if (ee == hipSuccess) {
            DeviceGuard gr(rp->device);
            ee = hipStreamWaitEvent(user, pl->shard_ev[1], 0);
        }
        pl->shard_used = true;
        if (e == hipSuccess) e = ee;
    }
    if (rc != INTFFT_OK) return rc;
    if (nc != 0) {
        rccl_report(nc_what, nc);
        return INTFFT_ERR_TRANSPORT;
    }
    return e == hipSuccess ? INTFFT_OK : (int)e;
}

static bool shard_uses_rccl(intfft_plan *const *plans, int nplans)
{
    bool use = true; // the whole set carries live communicators of this very set (one ncclCommInitAll: same set id, same owner)
    for (int i = 0; i < nplans; ++i)
        use = use && plans[i]->rccl_comm && plans[i]->rccl_rank == i && plans[i]->rccl_nranks == nplans && plans[i]->rccl_set != 0 &&
              plans[i]->rccl_set == plans[0]->rccl_set && plans[i]->rccl_owner == plans[0]->rccl_owner;
    return use;
}

int intfft_exec_sharded_async(intfft_plan *const *plans, int nplans, int root, const void *d_in, void *d_out, size_t batch, void *hip_stream)
{
    const int rc = shard_check(plans, nplans, root);
    if (rc != INTFFT_OK) return rc;
    if (batch && (!d_in || !d_out)) return INTFFT_ERR_NULL;
    if (batch == 0) return INTFFT_OK;
    return exec_sharded_enqueue(plans, nplans, root, d_in, d_out, batch, reinterpret_cast<hipStream_t>(hip_stream), shard_uses_rccl(plans, nplans));
}

int intfft_exec_sharded(intfft_plan *const *plans, int nplans, int root, const void *d_in, void *d_out, size_t batch)
{
    int rc = shard_check(plans, nplans, root);
    if (rc != INTFFT_OK) return rc;
    if (batch && (!d_in || !d_out)) return INTFFT_ERR_NULL;
    if (batch == 0) return INTFFT_OK;
    intfft_plan *rp = plans[root];
    hipError_t e = hipSuccess;
    {   // contract (intfft.h): d_in is complete once every stream of the root device is idle
        DeviceGuard g(rp->device);
        if (!g.ok) return INTFFT_ERR_NO_DEVICE;
        e = hipDeviceSynchronize();
        if (e == hipSuccess && !rp->s_call) e = hipStreamCreateWithFlags(&rp->s_call, hipStreamNonBlocking);
        if (e != hipSuccess) return (int)e;
    }
    rc = exec_sharded_enqueue(plans, nplans, root, d_in, d_out, batch, rp->s_call, shard_uses_rccl(plans, nplans));
    {
        DeviceGuard g(rp->device);
        e = hipStreamSynchronize(rp->s_call);
    }
    // The blocking entry point drains every stream the call used on every device, error or not: s_call already waits for all of them
    // through cross-device events, so on a healthy run these return at once -- a safety net for the multi-device orderings (the RCCL
    // transport in particular), which no box this library was measured on could exercise with two or more devices.
    for (int i = 0; i < nplans; ++i) {
        DeviceGuard g(plans[i]->device);
        for (hipStream_t st : {plans[i]->s_shard, plans[i]->s_shard2})
            if (st) {
                const hipError_t es = hipStreamSynchronize(st);
                if (e == hipSuccess && es != hipSuccess) e = es;
            }
    }
    if (rc != INTFFT_OK) return rc;
    return e == hipSuccess ? INTFFT_OK : (int)e;
}

int intfft_twiddles(const intfft_plan *plan, int stage, int32_t *h_out, size_t *count)
{
    if (!plan || !count) return INTFFT_ERR_NULL;
    if (stage == -1 && plan->l1) { // 2-D scheme: the inter-pass twiddles W_N^m, m = 0 .. N-1 (evaluated here, not stored)
        const size_t n = (size_t)1 << plan->L;
        *count = n;
        if (h_out)
            for (size_t m = 0; m < n; ++m) tw2d_eval(plan->L, plan->p.twdl_width, (unsigned)m, h_out[2 * m], h_out[2 * m + 1]);
        return INTFFT_OK;
    }
    const int nstages = plan->l1 ? std::max(plan->l1, plan->L - plan->l1) : plan->L;
    if (stage < 0 || stage >= nstages) return INTFFT_ERR_INVALID;
    const size_t n = (size_t)1 << stage;
    *count = n;
    if (h_out) std::memcpy(h_out, plan->h_tw.data() + (n - 1), n * sizeof(int2));
    return INTFFT_OK;
}

const char *intfft_strerror(int status)
{
    switch (status) {
    case INTFFT_OK: return "ok";
    case INTFFT_ERR_INVALID: return "invalid parameter";
    case INTFFT_ERR_UNSUPPORTED: return "unsupported configuration (the generic combination does not elaborate in the reference RTL, or the 2-D scheme does not cover it)";
    case INTFFT_ERR_TRANSPORT: return "shard transport unavailable or failed (RCCL not loadable, two plans on one device, or an RCCL call failed)";
    case INTFFT_ERR_NULL: return "null argument";
    case INTFFT_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case INTFFT_ERR_ALLOC: return "host allocation failed";
    default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

// intfft_version(): intfft_version.hip (its own translation unit: it carries the hash of the sources the library was built from)

} // extern "C"

namespace intfft {
void plan_geometry(const intfft_plan *plan, int *device, int *log2n, int *in_cb, int *out_cb)
{
    *device = plan->device, *log2n = plan->L, *in_cb = plan->in_cb, *out_cb = plan->out_cb;
}
} // namespace intfft

