// intfft_fastsmall.hip -- packed-int16 kernel for the shortest frames, N = 8, 16, 32: int_fftNk / int_ifftNk /
// int_fft_ifft_pair with NFFT = 3..5, DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled (the forward core in truncate or round
// mode, the inverse and the pair in truncate mode), natural order in and out.
//
// A whole frame fits the registers of ONE lane (N packed (re | im << 16) words), so every butterfly of every stage is
// lane-local, the bit reversal is a compile-time register renaming, and every twiddle index (position mod 2^s) is a
// compile-time register property: all twiddles are wave-uniform kernel arguments (SGPRs).  No lane exchange; LDS only to
// coalesce the global accesses (a wave moves 64 consecutive frames as full 1 KiB dwordx4 instructions).
// Same packed arithmetic as intfft_fast1024.hip (intfft_pk16.hpp), exact extraction (these frames are far from
// VALU-bound: 2.5 x fewer butterflies per byte than N = 1024).
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

struct SmallTw { // DIF packing {Wa, Wb} of STAGE 2, 3, 4 (table index = register index mod 2^s)
    u32 wa2[4], wb2[4], wa3[8], wb3[8], wa4[16], wb4[16];
};

enum { SM_FWD = 0, SM_INV = 1, SM_PAIR = 2 };

template <int L> __host__ __device__ constexpr int brev_small(int i)
{
    int r = 0;
    for (int b = 0; b < L; ++b) r |= ((i >> b) & 1) << (L - 1 - b);
    return r;
}
// butterfly b of STAGE S (b = 0 .. N/2 - 1): first register index and twiddle index
template <int S> __host__ __device__ constexpr int bf_reg(int b) { return ((b >> S) << (S + 1)) | (b & ((1 << S) - 1)); }

template <int L, int S, int ROUND>
__device__ __forceinline__ void small_dif_stage(u32 (&v)[1 << L], const u32 *wa_t, const u32 *wb_t, const Slice &sl)
{
    constexpr int N = 1 << L, H = 1 << S, M = H - 1;
#pragma unroll
    for (int g = 0; g < N / 2; g += 4) {
        const int i0 = bf_reg<S>(g), i1 = bf_reg<S>(g + 1), i2 = bf_reg<S>(g + 2), i3 = bf_reg<S>(g + 3);
        const u32 wa[4] = {wa_t[g & M], wa_t[(g + 1) & M], wa_t[(g + 2) & M], wa_t[(g + 3) & M]};
        const u32 wb[4] = {wb_t[g & M], wb_t[(g + 1) & M], wb_t[(g + 2) & M], wb_t[(g + 3) & M]};
        group4<ROUND, false, false, false, true, 0>(v[i0], v[i0 + H], v[i1], v[i1 + H], v[i2], v[i2 + H], v[i3], v[i3 + H], wa, wb, sl);
    }
}
template <int L, int S, int ROUND = 0>
__device__ __forceinline__ void small_dit_stage(u32 (&v)[1 << L], const u32 *wa_t, const u32 *wb_t, const Slice &sl)
{
    constexpr int N = 1 << L, H = 1 << S, M = H - 1;
#pragma unroll
    for (int g = 0; g < N / 2; g += 4) {
        const int i0 = bf_reg<S>(g), i1 = bf_reg<S>(g + 1), i2 = bf_reg<S>(g + 2), i3 = bf_reg<S>(g + 3);
        const u32 wa[4] = {wa_t[g & M], wa_t[(g + 1) & M], wa_t[(g + 2) & M], wa_t[(g + 3) & M]};
        const u32 wb[4] = {wb_t[g & M], wb_t[(g + 1) & M], wb_t[(g + 2) & M], wb_t[(g + 3) & M]};
        group4_dit<false, true, ROUND>(v[i0], v[i0 + H], v[i1], v[i1 + H], v[i2], v[i2 + H], v[i3], v[i3 + H], wa, wb, sl);
    }
}

template <int L, int ROUND> __device__ __forceinline__ void small_dif(u32 (&v)[1 << L], const SmallTw &t, const Slice &sl)
{
    constexpr int N = 1 << L;
    if constexpr (L >= 5) small_dif_stage<L, 4, ROUND>(v, t.wa4, t.wb4, sl);
    if constexpr (L >= 4) small_dif_stage<L, 3, ROUND>(v, t.wa3, t.wb3, sl);
    small_dif_stage<L, 2, ROUND>(v, t.wa2, t.wb2, sl);
    if constexpr (ROUND) {
        round_stages10<N, ROUND == 2>(v, sl);
        return;
    }
#pragma unroll
    for (int g = 0; g < N; g += 4) { // STAGE 1: even positions Y = D, odd positions Y = -j D (negation quirk)
        bfly_triv<false, false>(v[g], v[g + 2]);
        bfly_mj<false, false>(v[g + 1], v[g + 3]);
    }
#pragma unroll
    for (int g = 0; g < N; g += 2) bfly_triv<false, false>(v[g], v[g + 1]); // STAGE 0
}
template <int L, int ROUND = 0> __device__ __forceinline__ void small_dit(u32 (&v)[1 << L], const SmallTw &t, const Slice &sl)
{
    constexpr int N = 1 << L;
#pragma unroll
    for (int g = 0; g < N; g += 2) bfly_triv<(ROUND != 0), false>(v[g], v[g + 1]); // STAGE 0: T = B
    if constexpr (ROUND == 2) { // round mode on narrow data: the w-bit wrap of the rhu2 differences (intfft_pk16.hpp)
#pragma unroll
        for (int g = 1; g < N; g += 2) v[g] = wrap_w(v[g], sl.wd);
    }
#pragma unroll
    for (int g = 0; g < N; g += 4) { // STAGE 1
        bfly_triv<(ROUND != 0), false>(v[g], v[g + 2]);
        bfly_pj_dit<(ROUND != 0)>(v[g + 1], v[g + 3]);
    }
    if constexpr (ROUND == 2) {
#pragma unroll
        for (int g = 0; g < N; g += 4) v[g + 2] = wrap_w(v[g + 2], sl.wd), v[g + 3] = wrap_w(v[g + 3], sl.wd);
    }
    small_dit_stage<L, 2, ROUND>(v, t.wa2, t.wb2, sl);
    if constexpr (L >= 4) small_dit_stage<L, 3, ROUND>(v, t.wa3, t.wb3, sl);
    if constexpr (L >= 5) small_dit_stage<L, 4, ROUND>(v, t.wa4, t.wb4, sl);
}

// Global access is coalesced through a wave-private LDS tile: every dwordx4 load / store instruction of a wave covers
// 1 KiB of consecutive memory (lane l takes bytes 16 l of the instruction's KiB); the tile [64 frames][N + 4 dwords] is
// written row-major as loaded and each lane reads back its own frame (and the reverse for the stores).  Letting each
// lane access its own 4 N contiguous bytes directly thrashed the L1: N = 32 ran at half the speed of the generic kernel.
template <int L, int MODE, int ROUND>
__global__ __launch_bounds__(256) void k_fftsmall_i16(const u32 *in, u32 *out, const SmallTw t, size_t nframes, const Slice sl)
{
    constexpr int N = 1 << L, ROW = N + 4; // dwords per LDS row (16-byte aligned rows, padded against bank conflicts)
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * 64 * ROW];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *lds = lds_all + wv * 64 * ROW;
    const size_t nchunks = (nframes + 63) / 64; // 64 consecutive frames per wave pass
    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t ch = wave0; ch < nchunks; ch += nwaves) {
        const size_t f0 = ch * 64;
        const v4u *src = reinterpret_cast<const v4u *>(in + f0 * N) + lane;
        // coalesced loads -> LDS rows
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const int d = 256 * q + 4 * lane; // dword offset within the chunk
            const int fr = d >> L, e = d & (N - 1);
            v4u x = {0u, 0u, 0u, 0u};
            if (f0 + (size_t)fr < nframes) x = INTFFT_LD(src + 64 * q);
            *reinterpret_cast<v4u *>(lds + fr * ROW + e) = x;
        }
        wave_lds_fence(); // wave-private tile: LDS operations of one wave execute in order
        u32 v[N];
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const v4u x = *reinterpret_cast<const v4u *>(lds + lane * ROW + 4 * q);
            // the inverse core expects X[brev(n)] at position n (natural-order input): rename while reading
            if (MODE == SM_INV) {
                v[brev_small<L>(4 * q)] = x.x, v[brev_small<L>(4 * q + 1)] = x.y;
                v[brev_small<L>(4 * q + 2)] = x.z, v[brev_small<L>(4 * q + 3)] = x.w;
            } else {
                v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
            }
        }
        if (sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16: containers wrapped to w bits (exact w-bit extraction below)
        if (MODE != SM_INV) small_dif<L, ROUND>(v, t, sl);
        if (MODE != SM_FWD) small_dit<L, ROUND>(v, t, sl);
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            v4u y;
            if (MODE == SM_FWD) // natural-order output: X[k] sits at position brev(k)
                y = v4u{v[brev_small<L>(4 * q)], v[brev_small<L>(4 * q + 1)], v[brev_small<L>(4 * q + 2)], v[brev_small<L>(4 * q + 3)]};
            else
                y = v4u{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            *reinterpret_cast<v4u *>(lds + lane * ROW + 4 * q) = y;
        }
        wave_lds_fence();
        v4u *dst = reinterpret_cast<v4u *>(out + f0 * N) + lane;
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const int d = 256 * q + 4 * lane;
            const int fr = d >> L, e = d & (N - 1);
            const v4u y = *reinterpret_cast<const v4u *>(lds + fr * ROW + e);
            if (f0 + (size_t)fr < nframes) __builtin_nontemporal_store(y, dst + 64 * q);
        }
        wave_lds_fence();
    }
}

bool fastsmall_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly,
                         int in_order, int out_order)
{
    return log2n >= 3 && log2n <= 5 && packed_width_ok(data_width, format, rndmode) && twdl_width >= 8 && twdl_width <= 16 && format == 0 && use_fly == 1 &&
           in_order == 0 && out_order == 0;
}

const char *fastsmall_kernel_name() { return "k_fftsmall_i16"; }

template <int L, int MODE, int ROUND>
static hipError_t launch_sm(const u32 *in, u32 *out, const SmallTw &t, size_t nframes, const Slice &sl, hipStream_t stream)
{
    const size_t cap = resident_blocks(kptr(k_fftsmall_i16<L, MODE, ROUND>), 256, 4);
    const size_t need = ((nframes + 63) / 64 + 3) / 4;
    hipLaunchKernelGGL((k_fftsmall_i16<L, MODE, ROUND>), dim3((unsigned)(need < cap ? need : cap)), dim3(256), 0, stream, in, out, t,
                       nframes, sl);
    return hipGetLastError();
}

template <int L>
static hipError_t launch_sm_l(int direction, bool round, const u32 *in, u32 *out, const SmallTw &t, size_t nframes, const Slice &sl,
                              hipStream_t stream)
{
    const int rd = round ? (sl.wd != 16 ? 2 : 1) : 0; // round mode on narrow data: its own instantiation (the w-bit wraps)
#define INTFFT_SM(MODE)                                                                                          \
    return rd == 2 ? launch_sm<L, MODE, 2>(in, out, t, nframes, sl, stream)                                      \
                   : rd == 1 ? launch_sm<L, MODE, 1>(in, out, t, nframes, sl, stream) : launch_sm<L, MODE, 0>(in, out, t, nframes, sl, stream);
    if (direction == 1) { INTFFT_SM(SM_INV) }
    if (direction == 2) { INTFFT_SM(SM_PAIR) }
    INTFFT_SM(SM_FWD)
#undef INTFFT_SM
}

hipError_t launch_fastsmall(int log2n, int direction, int rnd_round, int twd, const void *in, void *out, const int2 *h_tw,
                            size_t nframes, hipStream_t stream, int data_width)
{
    if (nframes == 0) return hipSuccess;
    SmallTw t{};
    auto pack = [&](int s, u32 *wa, u32 *wb) {
        if (s >= log2n) return;
        for (int k = 0; k < (1 << s); ++k) {
            const int2 w = h_tw[(1 << s) - 1 + k];
            wa[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
            wb[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
        }
    };
    pack(2, t.wa2, t.wb2);
    pack(3, t.wa3, t.wb3);
    pack(4, t.wa4, t.wb4);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    switch (log2n) {
    case 3: return launch_sm_l<3>(direction, rnd_round != 0, pin, pout, t, nframes, sl, stream);
    case 4: return launch_sm_l<4>(direction, rnd_round != 0, pin, pout, t, nframes, sl, stream);
    default: return launch_sm_l<5>(direction, rnd_round != 0, pin, pout, t, nframes, sl, stream);
    }
}

} // namespace intfft
