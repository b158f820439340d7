// intfft_bigwlong.hip -- the general-width int32 class of intfft_bigw.hip at N = 2^17 .. 2^20 (round 5): int_fftNk with any DATA_WIDTH / TWDL_WIDTH / FORMAT /
// RNDMODE whose widths stay within 32 bits (18- or 24-bit scaled data, 32-bit scaled data, narrow unscaled data), natural order or the cores' own beat orders.  These lengths
// ran the generic k_pass<int32> passes (44-60 Gsample/s); intfft_bigw.hip stops at N = 2^16.
//
// N = 2^LX = B blocks of 2^16 points, B = 2^XS (int_fftNk.vhd:184-342: the DIF stages run STAGE LX-1 .. 0):
//   pass 0  k_bigw_pre<XS>   STAGE LX-1 .. 16 (over the block number b): user array -> plan scratch (int32 pairs at the core index)        -- this file
//   pass 1  k_bigw_a<16>     STAGE 15 .. 8 of every block, IN PLACE on the scratch (a workgroup reads its 256-row x 32-column tile whole, then writes
//                            the same tile: intfft_bigw.hip)
//   pass 2  k_bigw_b         STAGE 7 .. 0 + the natural-order store: its tiles are indexed by the TOP five bits of n at any length, so the kernel
//                            takes L = LX as it is
// (launch_bigw_m / launch_bigw_inv in intfft_bigw.hip string them together; the inverse mirrors it: k_bigw_qb at L = LX, k_bigw_qa<16> in place, k_bigw_post.)  Pass traffic: 8 + 8, 8 + 8, 8 + 8 = 48 B/sample for int32 containers against 16 algorithmic.
// Twiddles of pass 0: table entry 2^s - 1 + (n mod 2^s) of STAGE s, frame invariant, held in VGPRs over the workgroup's frame loop.
#define INTFFT_NT_LOADS 1
#include "intfft_u32.hpp"

namespace intfft {

// thread = P positions p = 256 P tile + 256 i + tid (i < P, P B = 16), registers [i][b]; grid = (frame groups) x (256 / P tiles)
template <int XS, int MODE, bool MASKED>
__global__ __launch_bounds__(256) void k_bigw_pre(const void *in, int2 *scr, const int2 *__restrict__ twt, const W32Args a, size_t nframes)
{
    static_assert(XS >= 1 && XS <= 4, "N = 2^17 .. 2^20");
    constexpr int LX = 16 + XS, B = 1 << XS, P = 16 >> XS, TILES = 256 / P;
    const int tid = threadIdx.x;
    const unsigned tile = blockIdx.x % TILES;
    const unsigned p0 = 256u * P * tile + (unsigned)tid;
    // stage ii (STAGE LX-1-ii) pairs (b, b + H), H = B >> (ii + 1): twiddle index (b mod H) 65536 + p of the table at 65536 H - 1
    int wr[P][B - 1], wi[P][B - 1]; // [i][H - 1 + j]
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int H = 1; H < B; H <<= 1)
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int2 w = twt[(size_t)65536 * H - 1 + (size_t)65536 * j + p0 + 256u * i];
                wr[i][H - 1 + j] = w.x, wi[i][H - 1 + j] = w.y;
            }
    typedef int v2i __attribute__((ext_vector_type(2)));
    const size_t fstep = gridDim.x / TILES;
    for (size_t f = blockIdx.x / TILES; f < nframes; f += fstep) {
        int re[16], im[16];
        unsigned toff = p0;
        asm volatile("" : "+v"(toff)); // opaque per iteration: the zero-extended offset stays one VGPR (intfft_device.hpp, at32)
        if (a.native & 1) { // HALVES order in (int_fftNk.vhd:15-21): beat (x[i], x[i + N/2]) = the blocks (b, b + B/2) at one position: adjacent samples, ONE access
            if (a.in16) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                const v2u *src = static_cast<const v2u *>(in) + (f << (LX - 1));
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int b = 0; b < B / 2; ++b) {
                        const v2u x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                        re[i * B + b] = (int)(x.x << a.in_sh) >> a.in_sh, im[i * B + b] = (int)(x.x << (a.in_sh - 16)) >> a.in_sh;
                        re[i * B + b + B / 2] = (int)(x.y << a.in_sh) >> a.in_sh, im[i * B + b + B / 2] = (int)(x.y << (a.in_sh - 16)) >> a.in_sh;
                    }
            } else {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i *src = static_cast<const v4i *>(in) + (f << (LX - 1));
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int b = 0; b < B / 2; ++b) {
                        const v4i x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                        re[i * B + b] = (int)((u32)x.x << a.in_sh) >> a.in_sh, im[i * B + b] = (int)((u32)x.y << a.in_sh) >> a.in_sh;
                        re[i * B + b + B / 2] = (int)((u32)x.z << a.in_sh) >> a.in_sh, im[i * B + b + B / 2] = (int)((u32)x.w << a.in_sh) >> a.in_sh;
                    }
            }
        } else if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + (f << LX);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const u32 x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                    re[i * B + b] = (int)(x << a.in_sh) >> a.in_sh; // conv_std_logic_vector(.., DATA_WIDTH): wrap on load
                    im[i * B + b] = (int)(x << (a.in_sh - 16)) >> a.in_sh;
                }
        } else {
            const v2i *src = static_cast<const v2i *>(in) + (f << LX);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const v2i x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                    re[i * B + b] = (int)((u32)x.x << a.in_sh) >> a.in_sh;
                    im[i * B + b] = (int)((u32)x.y << a.in_sh) >> a.in_sh;
                }
        }
#pragma unroll
        for (int ii = 0; ii < XS; ++ii) {
            const int H = B >> (ii + 1);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int g = 0; g < B; g += 2 * H)
#pragma unroll
                    for (int j = 0; j < H; ++j)
                        gfly<MODE, false, MASKED>(re[i * B + g + j], im[i * B + g + j], re[i * B + g + j + H], im[i * B + g + j + H], wr[i][H - 1 + j], wi[i][H - 1 + j],
                                                  a.st[LX - 1 - ii]);
        }
        v2i *dst = reinterpret_cast<v2i *>(scr) + (f << LX);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const v2i y = {re[i * B + b], im[i * B + b]};
                *at32(dst + 65536 * b + 256 * i, toff) = y;
            }
    }
}

template <int XS, int MODE, bool MASKED>
static hipError_t launch_pre_x(const W32Args &a, const void *in, int2 *scr, const int2 *tw, size_t nframes, hipStream_t stream)
{
    constexpr int TILES = 256 / (16 >> XS);
    size_t g = resident_blocks(kptr(k_bigw_pre<XS, MODE, MASKED>), 256, 4) / TILES;
    if (g < 1) g = 1;
    if (g > nframes) g = nframes;
    hipLaunchKernelGGL((k_bigw_pre<XS, MODE, MASKED>), dim3((unsigned)(g * TILES)), dim3(256), 0, stream, in, scr, tw, a, nframes);
    return hipGetLastError();
}
template <int MODE, bool MASKED>
static hipError_t launch_pre_m(int log2n, const W32Args &a, const void *in, int2 *scr, const int2 *tw, size_t nframes, hipStream_t stream)
{
    switch (log2n) {
    case 17: return launch_pre_x<1, MODE, MASKED>(a, in, scr, tw, nframes, stream);
    case 18: return launch_pre_x<2, MODE, MASKED>(a, in, scr, tw, nframes, stream);
    case 19: return launch_pre_x<3, MODE, MASKED>(a, in, scr, tw, nframes, stream);
    default: return launch_pre_x<4, MODE, MASKED>(a, in, scr, tw, nframes, stream);
    }
}

hipError_t launch_bigw_pre(int log2n, int mode, const W32Args &a, const void *in, int2 *scr, const int2 *tw, size_t nframes, hipStream_t stream)
{
    if (a.masked) {
        switch (mode) {
        case W_TRUNC: return launch_pre_m<W_TRUNC, true>(log2n, a, in, scr, tw, nframes, stream);
        case W_ROUND: return launch_pre_m<W_ROUND, true>(log2n, a, in, scr, tw, nframes, stream);
        default: return launch_pre_m<W_UNSCALED, true>(log2n, a, in, scr, tw, nframes, stream);
        }
    }
    switch (mode) {
    case W_TRUNC: return launch_pre_m<W_TRUNC, false>(log2n, a, in, scr, tw, nframes, stream);
    case W_ROUND: return launch_pre_m<W_ROUND, false>(log2n, a, in, scr, tw, nframes, stream);
    default: return launch_pre_m<W_UNSCALED, false>(log2n, a, in, scr, tw, nframes, stream);
    }
}

// ---- the inverse (int_ifftNk.vhd:183-341: DIT STAGE 0 .. LX-1): k_bigw_qb at L = LX (bit-reversed gather + STAGE 0 .. 7), k_bigw_qa<16> in place on the blocks
// (STAGE 8 .. 15), then STAGE 16 .. LX-1 here: scratch -> user array in natural order.  Same thread / tile shape as k_bigw_pre.
template <int XS, int MODE, bool MASKED>
__global__ __launch_bounds__(256) void k_bigw_post(const int2 *scr, void *out, const int2 *__restrict__ twt, const W32Args a, size_t nframes)
{
    static_assert(XS >= 1 && XS <= 4, "N = 2^17 .. 2^20");
    constexpr int LX = 16 + XS, B = 1 << XS, P = 16 >> XS, TILES = 256 / P;
    const int tid = threadIdx.x;
    const unsigned tile = blockIdx.x % TILES;
    const unsigned p0 = 256u * P * tile + (unsigned)tid;
    int wr[P][B - 1], wi[P][B - 1]; // [i][H - 1 + j]: STAGE 16 + log2 H pairs (b, b + H), index (b mod H) 65536 + p
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int H = 1; H < B; H <<= 1)
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int2 w = twt[(size_t)65536 * H - 1 + (size_t)65536 * j + p0 + 256u * i];
                wr[i][H - 1 + j] = w.x, wi[i][H - 1 + j] = w.y;
            }
    typedef int v2i __attribute__((ext_vector_type(2)));
    const size_t fstep = gridDim.x / TILES;
    for (size_t f = blockIdx.x / TILES; f < nframes; f += fstep) {
        int re[16], im[16];
        unsigned toff = p0;
        asm volatile("" : "+v"(toff));
        const v2i *src = reinterpret_cast<const v2i *>(scr) + (f << LX);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const v2i x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                re[i * B + b] = x.x, im[i * B + b] = x.y;
            }
#pragma unroll
        for (int ii = 0; ii < XS; ++ii) {
            const int H = 1 << ii;
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int g = 0; g < B; g += 2 * H)
#pragma unroll
                    for (int j = 0; j < H; ++j)
                        gfly_dit<MODE, false, MASKED>(re[i * B + g + j], im[i * B + g + j], re[i * B + g + j + H], im[i * B + g + j + H], wr[i][H - 1 + j], wi[i][H - 1 + j],
                                                      a.st[16 + ii]);
        }
        if (a.native & 1) { // HALVES order out (int_ifftNk.vhd:15-21): the blocks (b, b + B/2) of one position leave as one access
            if (a.out16) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                v2u *dst = static_cast<v2u *>(out) + (f << (LX - 1));
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int b = 0; b < B / 2; ++b) {
                        const v2u y = {((u32)re[i * B + b] & 0xFFFFu) | ((u32)im[i * B + b] << 16), ((u32)re[i * B + b + B / 2] & 0xFFFFu) | ((u32)im[i * B + b + B / 2] << 16)};
                        __builtin_nontemporal_store(y, at32(dst + 65536 * b + 256 * i, toff));
                    }
            } else {
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i *dst = static_cast<v4i *>(out) + (f << (LX - 1));
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int b = 0; b < B / 2; ++b) {
                        const v4i y = {re[i * B + b], im[i * B + b], re[i * B + b + B / 2], im[i * B + b + B / 2]};
                        __builtin_nontemporal_store(y, at32(dst + 65536 * b + 256 * i, toff));
                    }
            }
        } else if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + (f << LX);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B; ++b) __builtin_nontemporal_store(((u32)re[i * B + b] & 0xFFFFu) | ((u32)im[i * B + b] << 16), at32(dst + 65536 * b + 256 * i, toff));
        } else {
            v2i *dst = static_cast<v2i *>(out) + (f << LX);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const v2i y = {re[i * B + b], im[i * B + b]};
                    __builtin_nontemporal_store(y, at32(dst + 65536 * b + 256 * i, toff));
                }
        }
    }
}

template <int XS, int MODE, bool MASKED>
static hipError_t launch_post_x(const W32Args &a, const int2 *scr, void *out, const int2 *tw, size_t nframes, hipStream_t stream)
{
    constexpr int TILES = 256 / (16 >> XS);
    size_t g = resident_blocks(kptr(k_bigw_post<XS, MODE, MASKED>), 256, 4) / TILES;
    if (g < 1) g = 1;
    if (g > nframes) g = nframes;
    hipLaunchKernelGGL((k_bigw_post<XS, MODE, MASKED>), dim3((unsigned)(g * TILES)), dim3(256), 0, stream, scr, out, tw, a, nframes);
    return hipGetLastError();
}
template <int MODE, bool MASKED>
static hipError_t launch_post_m(int log2n, const W32Args &a, const int2 *scr, void *out, const int2 *tw, size_t nframes, hipStream_t stream)
{
    switch (log2n) {
    case 17: return launch_post_x<1, MODE, MASKED>(a, scr, out, tw, nframes, stream);
    case 18: return launch_post_x<2, MODE, MASKED>(a, scr, out, tw, nframes, stream);
    case 19: return launch_post_x<3, MODE, MASKED>(a, scr, out, tw, nframes, stream);
    default: return launch_post_x<4, MODE, MASKED>(a, scr, out, tw, nframes, stream);
    }
}

hipError_t launch_bigw_post(int log2n, int mode, const W32Args &a, const int2 *scr, void *out, const int2 *tw, size_t nframes, hipStream_t stream)
{
    if (a.masked) {
        switch (mode) {
        case W_TRUNC: return launch_post_m<W_TRUNC, true>(log2n, a, scr, out, tw, nframes, stream);
        case W_ROUND: return launch_post_m<W_ROUND, true>(log2n, a, scr, out, tw, nframes, stream);
        default: return launch_post_m<W_UNSCALED, true>(log2n, a, scr, out, tw, nframes, stream);
        }
    }
    switch (mode) {
    case W_TRUNC: return launch_post_m<W_TRUNC, false>(log2n, a, scr, out, tw, nframes, stream);
    case W_ROUND: return launch_post_m<W_ROUND, false>(log2n, a, scr, out, tw, nframes, stream);
    default: return launch_post_m<W_UNSCALED, false>(log2n, a, scr, out, tw, nframes, stream);
    }
}

bool bigw_long_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order)
{
    return log2n >= 17 && log2n <= 20 && data_width >= 2 && data_width + format * log2n <= 32 && twdl_width >= 4 && twdl_width <= 26 && use_fly == 1 &&
           (direction == 0 ? (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1)                      // int_fftNk: NATURAL | HALVES in, NATURAL | BITREV out
                           : direction == 1 && (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2));  // int_ifftNk: NATURAL | BITREV in, NATURAL | HALVES out
}

} // namespace intfft
