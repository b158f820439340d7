// intfft_fast4096.hip -- packed-int16 block kernel for N = 4096 (BASELINE config 5 and its two halves):
// int_fftNk / int_ifftNk / int_fft_ifft_pair with NFFT = 12, DATA_WIDTH = 16, TWDL_WIDTH <= 16,
// scaled-truncate, natural-order input and output (src/vhdl/main/int_fft_ifft_pair.vhd:161-330).
//
// One 256-thread workgroup owns one frame: 16 packed (re | im << 16) samples per thread, persistent loop
// over frames.  Twelve radix-2 stages = three in-register rounds of four stages, with a block-wide LDS
// transpose between rounds (two alternating 20 KiB regions -> one barrier per transpose):
//
//   layout LA  reg = n11..8, thread = n7..0                       DIF 11,10,9,8   / DIT 8,9,10,11
//   layout LB  reg = n7..4,  thread = (n11..8, n3..0)             DIF 7,6,5,4     / DIT 4,5,6,7
//   layout LC  reg = n3..0,  thread = rev8(n11..4)                DIF 3,2,1,0     / DIT 0,1,2,3
//
// The pair never leaves LC between the forward and the inverse core: after the DIF stages position n
// holds X[bitrev(n)], which is exactly the element int_ifftNk expects at position n -- the RTL wires the
// FFT lane outputs straight into the IFFT lane inputs (int_fft_ifft_pair.vhd:242-280).  The forward
// core alone stores from LC with the bit reversal folded into the thread mapping (every store
// instruction writes 256 contiguous bytes per wave); the inverse core alone loads into LC the same way.
//
// Round-A/B twiddles are frame invariant and live in VGPRs in the DIF packing {Wa, Wb}; the DIT
// butterfly reuses them by feeding the multiplier re/im-swapped, exactly like int_dit2_fly.vhd:304-322:
//   Bs = (B.im, B.re):   T.re = dot(Bs, Wb) = B.re*wr + B.im*wi,   T.im = dot(Bs, Wa) = B.im*wr - B.re*wi.
// Arithmetic, asm blocks, pre-shifted outputs and the guard-bit fast extraction: intfft_pk16.hpp /
// intfft_fast1024.hip.  The guard-bit bound covers the pair: through a scaled DIT stage the complex
// magnitude also grows by <= 1.42, so over 24 stages M <= 23172 + 35.
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

struct Fast4096Consts {
    u32 wa3[8], wb3[8]; // STAGE 3: table index r & 7
    u32 wa2[4], wb2[4]; // STAGE 2: table index r & 3
};

// frame-invariant per-thread twiddles of one round: stage with register offset 8 / 4 / 2 / 1
struct RoundTw {
    u32 wa8[8], wb8[8], wa4[4], wb4[4], wa2[2], wb2[2], wa1[1], wb1[1];
};

constexpr int ROW4K = 20;             // LDS row stride in dwords (16 data + 4 pad)
constexpr int REGION4K = 256 * ROW4K; // dwords per transpose region

__device__ __forceinline__ constexpr int rev4c(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// ---- DIT group of four general butterflies: a_i <- X, b_i <- Y (int_dit2_fly.vhd:142-162, 290-325) ----
template <bool FASTX, bool SG>
__device__ __forceinline__ void group4_dit(u32 &a0, u32 &b0, u32 &a1, u32 &b1, u32 &a2, u32 &b2, u32 &a3, u32 &b3,
                                           const u32 (&wa)[4], const u32 (&wb)[4], const Slice &sl)
{
    const u32 bs[4] = {__builtin_amdgcn_alignbit(b0, b0, 16), __builtin_amdgcn_alignbit(b1, b1, 16),
                       __builtin_amdgcn_alignbit(b2, b2, 16), __builtin_amdgcn_alignbit(b3, b3, 16)};
    u32 t[4]; // T >> 1
    if (FASTX) {
        mul4f<SG>(bs, bs, wb, wa, sl.sel_hi, t);
    } else {
        mul2x<15, SG>(bs[0], bs[0], wb[0], wa[0], bs[1], bs[1], wb[1], wa[1], sl.off_y1, sl.sel, t[0], t[1]);
        mul2x<15, SG>(bs[2], bs[2], wb[2], wa[2], bs[3], bs[3], wb[3], wa[3], sl.off_y1, sl.sel, t[2], t[3]);
    }
    const v2s A0 = as_v2s(a0) >> (short)1, A1 = as_v2s(a1) >> (short)1, A2 = as_v2s(a2) >> (short)1,
              A3 = as_v2s(a3) >> (short)1;
    a0 = as_u32(A0 + as_v2s(t[0]));
    b0 = as_u32(A0 - as_v2s(t[0]));
    a1 = as_u32(A1 + as_v2s(t[1]));
    b1 = as_u32(A1 - as_v2s(t[1]));
    a2 = as_u32(A2 + as_v2s(t[2]));
    b2 = as_u32(A2 - as_v2s(t[2]));
    a3 = as_u32(A3 + as_v2s(t[3]));
    b3 = as_u32(A3 - as_v2s(t[3]));
}

// DIT STAGE 1, odd positions: T.im = B.re, T.re = B.im >= 0 ? -B.im : ~B.im (int_dit2_fly.vhd:264-276)
__device__ __forceinline__ void bfly_pj_dit(u32 &a, u32 &b)
{
    const u32 rot = __builtin_amdgcn_alignbit(b, b, 16); // lo = B.im, hi = B.re
    const u32 nx = rot ^ 0x0000FFFFu;                     // lo = ~B.im
    const v2s add = {(short)((nx >> 15) & 1u), 0};        // + 1 in the low half iff B.im >= 0
    const u32 t = as_u32(as_v2s(nx) + add);
    sumdiff<false, false>(a, t, a, b);
}

// ---- four DIF stages on register offsets 8, 4, 2, 1 (stage numbers s0+3 .. s0) -----------------------
// kinds: inputs of the first stage are S-type (unshifted) unless VARSH0 gives a per-thread shift amount
template <bool FASTX, bool VARSH0>
__device__ __forceinline__ void dif_round(u32 (&v)[16], const RoundTw &tw, const Slice &sl, v2s shv)
{
    constexpr int M0 = 0, MA = 0xF;
    {
        const u32 wa0[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[2], tw.wa8[3]}, wb0[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[2], tw.wb8[3]};
        const u32 wa1[4] = {tw.wa8[4], tw.wa8[5], tw.wa8[6], tw.wa8[7]}, wb1[4] = {tw.wb8[4], tw.wb8[5], tw.wb8[6], tw.wb8[7]};
        group4<false, FASTX, false, true, false, M0, VARSH0>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl, shv);
        group4<false, FASTX, false, true, false, M0, VARSH0>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl, shv);
    }
    // offset 4: pairs (j, j+4); kind = j & 8
    group4<false, FASTX, false, true, false, M0>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], tw.wa4, tw.wb4, sl);
    group4<false, FASTX, false, true, false, MA>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], tw.wa4, tw.wb4, sl);
    // offset 2: pairs (j, j+2); twiddle j & 1; kind = j & 4
    {
        const u32 wa[4] = {tw.wa2[0], tw.wa2[1], tw.wa2[0], tw.wa2[1]}, wb[4] = {tw.wb2[0], tw.wb2[1], tw.wb2[0], tw.wb2[1]};
        group4<false, FASTX, false, true, false, M0>(v[0], v[2], v[1], v[3], v[8], v[10], v[9], v[11], wa, wb, sl);
        group4<false, FASTX, false, true, false, MA>(v[4], v[6], v[5], v[7], v[12], v[14], v[13], v[15], wa, wb, sl);
    }
    // offset 1: pairs (j, j+1); kind = j & 2
    {
        const u32 wa[4] = {tw.wa1[0], tw.wa1[0], tw.wa1[0], tw.wa1[0]}, wb[4] = {tw.wb1[0], tw.wb1[0], tw.wb1[0], tw.wb1[0]};
        group4<false, FASTX, false, true, false, M0>(v[0], v[1], v[4], v[5], v[8], v[9], v[12], v[13], wa, wb, sl);
        group4<false, FASTX, false, true, false, MA>(v[2], v[3], v[6], v[7], v[10], v[11], v[14], v[15], wa, wb, sl);
    }
}

// ---- four DIT stages on register offsets 1, 2, 4, 8 --------------------------------------------------
template <bool FASTX>
__device__ __forceinline__ void dit_round(u32 (&v)[16], const RoundTw &tw, const Slice &sl)
{
    {
        const u32 wa[4] = {tw.wa1[0], tw.wa1[0], tw.wa1[0], tw.wa1[0]}, wb[4] = {tw.wb1[0], tw.wb1[0], tw.wb1[0], tw.wb1[0]};
        group4_dit<FASTX, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], wa, wb, sl);
        group4_dit<FASTX, false>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], wa, wb, sl);
    }
    {
        const u32 wa[4] = {tw.wa2[0], tw.wa2[1], tw.wa2[0], tw.wa2[1]}, wb[4] = {tw.wb2[0], tw.wb2[1], tw.wb2[0], tw.wb2[1]};
        group4_dit<FASTX, false>(v[0], v[2], v[1], v[3], v[4], v[6], v[5], v[7], wa, wb, sl);
        group4_dit<FASTX, false>(v[8], v[10], v[9], v[11], v[12], v[14], v[13], v[15], wa, wb, sl);
    }
    group4_dit<FASTX, false>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], tw.wa4, tw.wb4, sl);
    group4_dit<FASTX, false>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], tw.wa4, tw.wb4, sl);
    {
        const u32 wa0[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[2], tw.wa8[3]}, wb0[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[2], tw.wb8[3]};
        const u32 wa1[4] = {tw.wa8[4], tw.wa8[5], tw.wa8[6], tw.wa8[7]}, wb1[4] = {tw.wb8[4], tw.wb8[5], tw.wb8[6], tw.wb8[7]};
        group4_dit<FASTX, false>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
        group4_dit<FASTX, false>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
    }
}

// ---- round C: DIF stages 3,2,1,0 / DIT stages 0,1,2,3 on reg = n3..0, uniform twiddles ----------------
template <bool FASTX> __device__ __forceinline__ void dif_round_c(u32 (&v)[16], const Fast4096Consts &c, const Slice &sl, v2s shv)
{
    const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    group4<false, FASTX, false, true, true, 0, true>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl, shv);
    group4<false, FASTX, false, true, true, 0, true>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl, shv);
    group4<false, FASTX, false, true, true, 0>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
    group4<false, FASTX, false, true, true, 0xF>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
#pragma unroll
    for (int g = 0; g < 16; g += 8) { // stage 1: kind = r & 4
        bfly_triv<false, false>(v[g], v[g + 2]);
        bfly_mj<false, false>(v[g + 1], v[g + 3]);
        bfly_triv<false, true>(v[g + 4], v[g + 6]);
        bfly_mj<false, true>(v[g + 5], v[g + 7]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) bfly_triv<false, false>(v[g], v[g + 1]);
}

template <bool FASTX> __device__ __forceinline__ void dit_round_c(u32 (&v)[16], const Fast4096Consts &c, const Slice &sl)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2) bfly_triv<false, false>(v[g], v[g + 1]); // STAGE 0: T = B
#pragma unroll
    for (int g = 0; g < 16; g += 4) { // STAGE 1: even positions T = B, odd positions T = +j B (quirk)
        bfly_triv<false, false>(v[g], v[g + 2]);
        bfly_pj_dit(v[g + 1], v[g + 3]);
    }
    group4_dit<FASTX, true>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
    group4_dit<FASTX, true>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
    const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    group4_dit<FASTX, true>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
    group4_dit<FASTX, true>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
}

// ---- block-wide transposes: write 16 scattered dwords, barrier, read one padded row ------------------
// OFF(j): compile-time row offset (in rows) of register j; wr: this thread's base (dword index)
#define INTFFT_X_READ(region)                                                                          \
    {                                                                                                  \
        __syncthreads();                                                                               \
        const uint4 *rp = reinterpret_cast<const uint4 *>((region) + ROW4K * tid);                     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
        {                                                                                              \
            const uint4 x = rp[q];                                                                     \
            v[4 * q + 0] = x.x;                                                                        \
            v[4 * q + 1] = x.y;                                                                        \
            v[4 * q + 2] = x.z;                                                                        \
            v[4 * q + 3] = x.w;                                                                        \
        }                                                                                              \
    }

enum { MODE_FWD = 0, MODE_INV = 1, MODE_PAIR = 2 };

template <int MODE, bool FAST_OK>
__global__ __launch_bounds__(256) void k_fft4096_i16(const u32 *in, u32 *out, const int2 *__restrict__ twt,
                                                     const Fast4096Consts c, size_t nframes, const Slice sl)
{
    __shared__ __attribute__((aligned(16))) u32 lds[2 * REGION4K];
    u32 *const reg0 = lds, *const reg1 = lds + REGION4K;
    volatile u32 *const s_unsafe = lds + (ROW4K - 1); // a pad cell of row 0 (columns 16..19 are never transposed)
    const int tid = threadIdx.x;
    const int lo4 = tid & 15, hi4 = tid >> 4;

    // ---- frame-invariant twiddles (DIF packing), stage s table at twt + 2^s - 1 ----
    // round A: thread = n7..0, regs n11..8 -> stage 11 index 256*jj + tid, ... stage 8 index tid
    // round B: thread low nibble q = n3..0, regs n7..4 -> stage 7 index 16*jj + q, ... stage 4 index q
    RoundTw ta, tb;
    auto ld = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = twt[idx];
        wa = pack_wa(w);
        wb = pack_wb(w);
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) ld(2047 + 256 * j + tid, ta.wa8[j], ta.wb8[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ld(1023 + 256 * j + tid, ta.wa4[j], ta.wb4[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) ld(511 + 256 * j + tid, ta.wa2[j], ta.wb2[j]);
    ld(255 + tid, ta.wa1[0], ta.wb1[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) ld(127 + 16 * j + lo4, tb.wa8[j], tb.wb8[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ld(63 + 16 * j + lo4, tb.wa4[j], tb.wb4[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) ld(31 + 16 * j + lo4, tb.wa2[j], tb.wb2[j]);
    ld(15 + lo4, tb.wa1[0], tb.wb1[0]);

    // ---- transpose addressing (dword offsets of this thread; register part is compile time) ----
    // LA -> LB and LB -> LA: element (thread x, reg y) -> row 16*y + x3..0, column x7..4
    const int w_ab = ROW4K * lo4 + hi4;
    // LB -> LC: thread t' = (n11..8, n3..0), reg j' = n7..4 -> row rev4(n11..8) + 16*rev4(j'), column n3..0
    const int rv_hi = ((hi4 & 1) << 3) | ((hi4 & 2) << 1) | ((hi4 & 4) >> 1) | ((hi4 & 8) >> 3);
    const int w_bc = ROW4K * rv_hi + lo4;
    // LC -> LB: thread t'' = rev8(n11..4), reg r = n3..0 -> row 16*rev4(t''3..0) + r, column rev4(t''7..4)
    const int rv_lo = ((lo4 & 1) << 3) | ((lo4 & 2) << 1) | ((lo4 & 4) >> 1) | ((lo4 & 8) >> 3);
    const int w_cb = ROW4K * 16 * rv_lo + rv_hi;
    // per-thread shift amounts where the value kind depends on a thread bit after a transpose
    const short shb = (short)(1 - (hi4 & 1));        // LB: kind = n8 = t'4
    const short shc = (short)(1 - ((tid >> 7) & 1)); // LC: kind = n4 = t''7
    const v2s sh_b = {shb, shb}, sh_c = {shc, shc};

    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        u32 v[16];
        const u32 *src = in + f * 4096;
        u32 *dst = out + f * 4096;
        if (MODE == MODE_INV) { // LC: v[r] = X[rev12(n)], rev12(n) = 256*rev4(r) + t''
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = __builtin_nontemporal_load(src + 256 * rev4c(r) + tid);
        } else { // LA: v[j] = x[256 j + tid]
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(src + 256 * j + tid);
        }

        // guard-bit test of the whole frame (block-uniform)
        bool fast = false;
        if (FAST_OK) {
            if (tid == 0) *s_unsafe = 0;
            __syncthreads();
            if (guard_acc(v) != 0) *s_unsafe = 1;
            __syncthreads();
            fast = *s_unsafe == 0;
        }

#define INTFFT_BODY(FX)                                                                                 \
    {                                                                                                   \
        if (MODE != MODE_INV) {                                                                         \
            dif_round<FX, false>(v, ta, sl, sh_b);                                                      \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) reg0[w_ab + ROW4K * 16 * j] = v[j];          \
            INTFFT_X_READ(reg0)                                                                         \
            dif_round<FX, true>(v, tb, sl, sh_b);                                                       \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) reg1[w_bc + ROW4K * 16 * rev4c(j)] = v[j];   \
            INTFFT_X_READ(reg1)                                                                         \
            dif_round_c<FX>(v, c, sl, sh_c);                                                            \
        }                                                                                               \
        if (MODE == MODE_FWD) {                                                                         \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                              \
                __builtin_nontemporal_store(v[r], dst + 256 * rev4c(r) + tid);                          \
        } else {                                                                                        \
            dit_round_c<FX>(v, c, sl);                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) reg0[w_cb + ROW4K * r] = v[r];               \
            INTFFT_X_READ(reg0)                                                                         \
            dit_round<FX>(v, tb, sl);                                                                   \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) reg1[w_ab + ROW4K * 16 * j] = v[j];          \
            INTFFT_X_READ(reg1)                                                                         \
            dit_round<FX>(v, ta, sl);                                                                   \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                              \
                __builtin_nontemporal_store(v[j], dst + 256 * j + tid);                                 \
        }                                                                                               \
    }
        if (FAST_OK && fast) INTFFT_BODY(FAST_OK)
        else INTFFT_BODY(false)
#undef INTFFT_BODY
        __syncthreads(); // region reuse by the next frame (and s_unsafe)
    }
}

bool fast4096_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int use_fly,
                        int in_order, int out_order)
{
    return log2n == 12 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 && rndmode == 0 &&
           use_fly == 1 && in_order == 0 && out_order == 0;
}

const char *fast4096_kernel_name() { return "k_fft4096_i16"; }

template <int MODE, bool FAST_OK>
static hipError_t launch4k(const u32 *in, u32 *out, const int2 *tw, const Fast4096Consts &c, size_t nframes,
                           const Slice &sl, hipStream_t stream)
{
    static int per_cu = 0, cus = 0;
    if (!per_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fft4096_i16<MODE, FAST_OK>, 256, 0) != hipSuccess ||
            per_cu <= 0)
            per_cu = 2;
        if (const char *e = getenv("INTFFT_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    }
    const size_t cap = (size_t)cus * (size_t)per_cu;
    const unsigned blocks = (unsigned)(nframes < cap ? nframes : cap);
    hipLaunchKernelGGL((k_fft4096_i16<MODE, FAST_OK>), dim3(blocks), dim3(256), 0, stream, in, out, tw, c, nframes, sl);
    return hipGetLastError();
}

hipError_t launch_fast4096(int direction, int twd, const void *in, void *out, const int2 *tw_all, const int2 *h_tw,
                           size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Fast4096Consts c;
    for (int k = 0; k < 8; ++k) {
        const int2 w = h_tw[7 + k];
        c.wa3[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb3[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    for (int k = 0; k < 4; ++k) {
        const int2 w = h_tw[3 + k];
        c.wa2[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb2[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    const Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = getenv("INTFFT_FAST_EXTRACT") ? atoi(getenv("INTFFT_FAST_EXTRACT")) : 1;
    const bool fast_ok = twd == 16 && allow_fast;
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    switch (direction) {
    case 0:
        return fast_ok ? launch4k<MODE_FWD, true>(pin, pout, tw_all, c, nframes, sl, stream)
                       : launch4k<MODE_FWD, false>(pin, pout, tw_all, c, nframes, sl, stream);
    case 1:
        return fast_ok ? launch4k<MODE_INV, true>(pin, pout, tw_all, c, nframes, sl, stream)
                       : launch4k<MODE_INV, false>(pin, pout, tw_all, c, nframes, sl, stream);
    default:
        return fast_ok ? launch4k<MODE_PAIR, true>(pin, pout, tw_all, c, nframes, sl, stream)
                       : launch4k<MODE_PAIR, false>(pin, pout, tw_all, c, nframes, sl, stream);
    }
}

} // namespace intfft
