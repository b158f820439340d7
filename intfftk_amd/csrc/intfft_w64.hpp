// intfft_w64.hpp -- the 64-bit wave kernels as templates (intfft_fastw64.hip: N = 1024 and the launcher; intfft_fastw64s.hip: N = 64 .. 512)
// 64-bit wave kernels: int_fftNk / int_ifftNk with NFFT = 6 .. 10 (N = 64 .. 1024), natural order in and out, for every DATA_WIDTH /
// TWDL_WIDTH / FORMAT / RNDMODE whose results need more than 32 and at most 64 bits -- 32-bit unscaled data (42-bit results,
// int_fft_single_path.vhd:15 documents DATA_WIDTH 8-32), 24-bit unscaled data with 16- or 24-bit twiddles, wide scaled data, the
// row cores of unscaled 2-D scheme plans.  These plans ran on the generic LDS pass kernel k_pass<int64> (37-55 Gsample/s).
//
// Same wave mapping as intfft_fastw32.hip / intfft_fast1024u.hip: one wave64 owns one frame, 16 complex samples per lane as 64-bit
// register pairs, stages 9..6 in registers, two lane swaps (v_permlane32_swap / v_permlane16_swap on both halves of a value) with
// stages 5 and 4, one wave-private LDS transpose (four dword planes: re.lo, re.hi, im.lo, im.hi), stages 3..0 in registers on
// wave-uniform twiddles, the bit reversal (int_bitrev_order.vhd:82-104) folded into the transpose so that a store instruction writes
// 1 KiB.  The arithmetic is the generic device form of the RTL (intfft_device.hpp: dif_fly<int64_t> with the planner's StageDesc
// -- every regime of int_cmult_dsp48.vhd:182-434 incl. the XSER-dependent ones, all three sum / difference variants of
// int_dif2_fly.vhd:144-241), with the rounding kind as a template parameter.
#pragma once
#include "intfft_u32.hpp"


namespace intfft {

using i64 = long long;

struct W64Args {
    StageDesc st[10]; // indexed by the STAGE generic
    int in_cb;        // input container bytes per component: 4 or 8
    int dw;           // DATA_WIDTH (inputs are wrapped to it on load: conv_std_logic_vector, fft_signle_test.vhd:163-164)
};

__device__ __forceinline__ void swap32_64(i64 &a, i64 &b)
{
    int al = (int)a, ah = (int)(a >> 32), bl = (int)b, bh = (int)(b >> 32);
    uswap32(al, bl);
    uswap32(ah, bh);
    a = (i64)(((unsigned long long)(u32)ah << 32) | (u32)al);
    b = (i64)(((unsigned long long)(u32)bh << 32) | (u32)bl);
}
__device__ __forceinline__ void swap16_64(i64 &a, i64 &b)
{
    int al = (int)a, ah = (int)(a >> 32), bl = (int)b, bh = (int)(b >> 32);
    uswap16(al, bl);
    uswap16(ah, bh);
    a = (i64)(((unsigned long long)(u32)ah << 32) | (u32)al);
    b = (i64)(((unsigned long long)(u32)bh << 32) | (u32)bl);
}

// CM: what the launcher knows about the multiplier stages of the plan -- 1: every one has StageDesc::narrow == 1 (mw + TWDL_WIDTH <= 64:
// each product fits one int64); 3: every one has mw <= 63 and a + b <= 31 (three-dword products, two v_mad_i64_i32 each: 24-bit
// twiddles on data beyond 40 bits); 0: nothing (regime test at run time, every form in the instruction stream: 240 VGPRs)
template <int RNDC, int CLS, int CM, bool DIT = false>
__device__ __forceinline__ void fly64(const StageDesc &st_in, int odd, i64 &are, i64 &aim, i64 &bre, i64 &bim, int wr, int wi)
{
    StageDesc st = st_in;
    if (CM == 1) st.narrow = 1;
    if (CM == 3) st.narrow = -3;
    Cx<int64_t> x, y;
    if (DIT) dit_fly<int64_t, RNDC, CLS>(st, odd, Cx<int64_t>{(int64_t)are, (int64_t)aim}, Cx<int64_t>{(int64_t)bre, (int64_t)bim}, wr, wi, x, y);
    else dif_fly<int64_t, RNDC, CLS>(st, odd, Cx<int64_t>{(int64_t)are, (int64_t)aim}, Cx<int64_t>{(int64_t)bre, (int64_t)bim}, wr, wi, x, y);
    are = x.re, aim = x.im, bre = y.re, bim = y.im;
    if (CLS == 2) __builtin_amdgcn_sched_barrier(0); // one multiplier butterfly at a time: interleaved, their 64-bit products set the VGPR count
}

constexpr int PLANE64 = 64 * ROWU; // dwords per transpose plane of one wave
#ifndef W64_WAVES
#define W64_WAVES 2
#endif

template <int L, int RNDC, int CM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W64_WAVES, W64_WAVES))) void k_fft1024_w64(const void *in, i64 *out, const int2 *__restrict__ twt, const UConsts c, const W64Args a,
                                                     size_t nframes_user)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[]; // 4 waves x 4 planes x 64 rows x ROWU
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *lds = lds_all + wv * 4 * PLANE64;
    constexpr int FP = 1 << (10 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks of 1024 samples

    int w9r[8] = {}, w9i[8] = {}, w8r[4] = {}, w8i[4] = {}, w7r[2] = {}, w7i[2] = {}, w6r = 0, w6i = 0, w5r, w5i, w4r, w4i;
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int2 w = twt[511 + 64 * j + lane];
            w9r[j] = w.x, w9i[j] = w.y;
        }
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int2 w = twt[255 + 64 * j + lane];
            w8r[j] = w.x, w8i[j] = w.y;
        }
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int2 w = twt[127 + 64 * j + lane];
            w7r[j] = w.x, w7i[j] = w.y;
        }
    }
    {
        int2 w;
        if constexpr (L >= 7) {
            w = twt[63 + lane];
            w6r = w.x, w6i = w.y;
        }
        w = twt[31 + (lane & 31)];
        w5r = w.x, w5i = w.y;
        w = twt[15 + (lane & 15)];
        w4r = w.x, w4i = w.y;
    }
    // after the two lane swaps: lane5 = a9, lane4 = a8, lane3..0 = a3..0; reg j3 = a5, j2 = a4, j1 = a7, j0 = a6 (intfft_fast1024.hip).
    // Destination row = new lane with lane bit lane_bit_u<L>(k) = a_k (N = 1024: lane bit i = a(9 - i), the output index is then
    // rev4(r) * 64 + row, contiguous in the row; shorter frames: one more lane swap before the store, intfft_u32.hpp)
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    u32 *wr_base = lds + ROWU * ((t5 << lane_bit_u<L>(9)) + (t4 << lane_bit_u<L>(8))) + (lane & 15);
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROWU * lane);
    constexpr int s7 = lane_bit_u<L>(7), s6 = lane_bit_u<L>(6), s5 = lane_bit_u<L>(5), s4 = lane_bit_u<L>(4);
    int lane_off = 0, lane_frame = 0;
    if constexpr (L < 10) {
        lane_off = ((lane >> 5) & 1) * out_weight<L>(3);
#pragma unroll
        for (int k = 4; k < 10; ++k) {
            if (k == L - 1) continue;
            const int bit = (lane >> lane_bit_u<L>(k)) & 1;
            lane_off += bit * out_weight<L>(k);
            if (k >= L) lane_frame += bit << (k - L);
        }
    }
    constexpr int ow0 = out_weight<L>(0), ow1 = out_weight<L>(1), ow2 = out_weight<L>(2);

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user; // last chunk: absent frames read as 0
        i64 re[16], im[16];
        if (a.in_cb == 4) {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + f * 1024 + lane;
            const int sh = 32 - a.dw;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v2i x = {0, 0};
                if (!partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) x = INTFFT_LD(src + 64 * j);
                re[j] = (int)((u32)x.x << sh) >> sh;
                im[j] = (int)((u32)x.y << sh) >> sh;
            }
        } else {
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            const v2l *src = static_cast<const v2l *>(in) + f * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v2l x = {0, 0};
                if (!partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) x = INTFFT_LD(src + 64 * j);
                re[j] = wrapw<int64_t>((int64_t)x.x, a.dw);
                im[j] = wrapw<int64_t>((int64_t)x.y, a.dw);
            }
        }
        // ---- stages 9..6 in registers (register offsets 8, 4, 2, 1) ----
        if constexpr (L >= 10) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fly64<RNDC, 2, CM>(a.st[9], 0, re[j], im[j], re[j + 8], im[j + 8], w9r[j], w9i[j]);
        }
        if constexpr (L >= 9) {
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int j = 0; j < 4; ++j) fly64<RNDC, 2, CM>(a.st[8], 0, re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w8r[j], w8i[j]);
        }
        if constexpr (L >= 8) {
#pragma unroll
            for (int g = 0; g < 16; g += 4)
#pragma unroll
                for (int j = 0; j < 2; ++j) fly64<RNDC, 2, CM>(a.st[7], 0, re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w7r[j], w7i[j]);
        }
        if constexpr (L >= 7) {
#pragma unroll
            for (int g = 0; g < 16; g += 2) fly64<RNDC, 2, CM>(a.st[6], 0, re[g], im[g], re[g + 1], im[g + 1], w6r, w6i);
        }
        // ---- lane bit 5 <-> reg bit 3, stage 5; lane bit 4 <-> reg bit 2, stage 4 ----
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            swap32_64(re[j], re[j + 8]);
            swap32_64(im[j], im[j + 8]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) fly64<RNDC, 2, CM>(a.st[5], 0, re[j], im[j], re[j + 8], im[j + 8], w5r, w5i);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                swap16_64(re[g + j], re[g + j + 4]);
                swap16_64(im[g + j], im[g + j + 4]);
            }
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) fly64<RNDC, 2, CM>(a.st[4], 0, re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r, w4i);
        // ---- LDS transpose: regs become a3..0 (four dword planes) ----
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;
            const int row_j = (j1 << s7) + (j0 << s6) + (j3 << s5) + (j2 << s4);
            wr_base[ROWU * row_j] = (u32)re[j];
            wr_base[PLANE64 + ROWU * row_j] = (u32)((unsigned long long)re[j] >> 32);
            wr_base[2 * PLANE64 + ROWU * row_j] = (u32)im[j];
            wr_base[3 * PLANE64 + ROWU * row_j] = (u32)((unsigned long long)im[j] >> 32);
        }
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd_base[q], xh = rd_base[q + PLANE64 / 4], y = rd_base[q + 2 * (PLANE64 / 4)], yh = rd_base[q + 3 * (PLANE64 / 4)];
            const u32 xl[4] = {x.x, x.y, x.z, x.w}, xu[4] = {xh.x, xh.y, xh.z, xh.w}, yl[4] = {y.x, y.y, y.z, y.w}, yu[4] = {yh.x, yh.y, yh.z, yh.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                re[4 * q + i] = (i64)(((unsigned long long)xu[i] << 32) | xl[i]);
                im[4 * q + i] = (i64)(((unsigned long long)yu[i] << 32) | yl[i]);
            }
        }
        wave_lds_fence();
        // ---- stages 3, 2 (wave-uniform twiddles), 1, 0 (multiplier-free) ----
#pragma unroll
        for (int r = 0; r < 8; ++r) fly64<RNDC, 2, CM>(a.st[3], 0, re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) fly64<RNDC, 2, CM>(a.st[2], 0, re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r]);
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            fly64<RNDC, 1, CM>(a.st[1], 0, re[g], im[g], re[g + 2], im[g + 2], 0, 0);
            fly64<RNDC, 1, CM>(a.st[1], 1, re[g + 1], im[g + 1], re[g + 3], im[g + 3], 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) fly64<RNDC, 1, CM>(a.st[0], 0, re[g], im[g], re[g + 1], im[g + 1], 0, 0);
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        if constexpr (L < 10) { // lane bit 5 <-> reg bit 3: every lane holds pairs of consecutive outputs (32 bytes)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                swap32_64(re[r], re[r + 8]);
                swap32_64(im[r], im[r + 8]);
            }
            if (f * FP + (size_t)lane_frame < nframes_user) {
                v2l *dst = reinterpret_cast<v2l *>(out) + f * 1024 + lane_off;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    v2l *d = dst + (q & 1) * ow0 + ((q >> 1) & 1) * ow1 + (q >> 2) * ow2;
                    const v2l y0 = {re[q], im[q]}, y1 = {re[q + 8], im[q + 8]};
                    __builtin_nontemporal_store(y0, d);
                    __builtin_nontemporal_store(y1, d + 1);
                }
            }
        } else { // X index = rev4(r) * 64 + lane
            v2l *dst = reinterpret_cast<v2l *>(out) + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                const v2l y = {re[r], im[r]};
                __builtin_nontemporal_store(y, dst + 64 * rr);
            }
        }
    }
}

// int_ifftNk (int_ifftNk.vhd:183-341), natural order in and out: the mirror.  Reg r <- X[lane + 64 rev4(r)] (position a9..4 = rev6(lane),
// a3..0 = r: the bit reversal of int_bitrev_order.vhd:82-104 in the addressing), DIT 0..3 in registers, LDS transpose to
// lane = (a9 a8 a3..0), reg = (a5 a4 a7 a6), DIT 4, lane bit 4 <-> reg bit 2, DIT 5, lane bit 5 <-> reg bit 3, DIT 6..9 on
// reg = a9..6, lane = a5..0; every store instruction writes 1 KiB of the natural-order result.
template <int L, int RNDC, int CM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W64_WAVES, W64_WAVES))) void k_ifft1024_w64(const void *in, i64 *out, const int2 *__restrict__ twt, const UConsts c, const W64Args a,
                                                      size_t nframes_user)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *lds = lds_all + wv * 4 * PLANE64;
    constexpr int FP = 1 << (10 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP;

    int w9r[8] = {}, w9i[8] = {}, w8r[4] = {}, w8i[4] = {}, w7r[2] = {}, w7i[2] = {}, w6r = 0, w6i = 0, w5r, w5i, w4r, w4i;
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int2 w = twt[511 + 64 * j + lane];
            w9r[j] = w.x, w9i[j] = w.y;
        }
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int2 w = twt[255 + 64 * j + lane];
            w8r[j] = w.x, w8i[j] = w.y;
        }
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int2 w = twt[127 + 64 * j + lane];
            w7r[j] = w.x, w7i[j] = w.y;
        }
    }
    {
        int2 w;
        if constexpr (L >= 7) {
            w = twt[63 + lane];
            w6r = w.x, w6i = w.y;
        }
        w = twt[31 + (lane & 31)];
        w5r = w.x, w5i = w.y;
        w = twt[15 + (lane & 15)];
        w4r = w.x, w4i = w.y;
    }
    // while loading, lane bit lane_bit_u<L>(k) = a_k (N = 1024: lane bit i = a(9 - i)); transpose: row = 32 a9 + 16 a8 + r, column = (a5 a4 a7 a6)
    auto ab = [&](int k) { return (lane >> lane_bit_u<L>(k)) & 1; };
    u32 *wr_inv = lds + ROWU * (32 * ab(9) + 16 * ab(8)) + ((ab(5) << 3) | (ab(4) << 2) | (ab(7) << 1) | ab(6));
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROWU * lane);

    // N < 1024: every lane loads pairs of consecutive X (mirror of the forward kernel's store); before the swap lane bit 5 = a3
    int lane_off = 0, lane_frame = 0;
    if constexpr (L < 10) {
        lane_off = ((lane >> 5) & 1) * out_weight<L>(3);
#pragma unroll
        for (int k = 4; k < 10; ++k) {
            if (k == L - 1) continue;
            lane_off += ab(k) * out_weight<L>(k);
            if (k >= L) lane_frame += ab(k) << (k - L);
        }
    }
    constexpr int ow0 = out_weight<L>(0), ow1 = out_weight<L>(1), ow2 = out_weight<L>(2);

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user;
        i64 re[16], im[16];
        if constexpr (L < 10) {
            const bool ok = !partial || f * FP + (size_t)lane_frame < nframes_user;
            if (a.in_cb == 4) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                typedef int v2i __attribute__((ext_vector_type(2)));
                const int sh = 32 - a.dw;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v2i *src = static_cast<const v2i *>(in) + f * 1024 + lane_off + (q & 1) * ow0 + ((q >> 1) & 1) * ow1 + (q >> 2) * ow2;
                    v4i x = {0, 0, 0, 0};
                    if (ok) x = INTFFT_LD(reinterpret_cast<const v4i *>(src));
                    re[q] = (int)((u32)x.x << sh) >> sh, im[q] = (int)((u32)x.y << sh) >> sh;
                    re[q + 8] = (int)((u32)x.z << sh) >> sh, im[q + 8] = (int)((u32)x.w << sh) >> sh;
                }
            } else {
                typedef i64 v2l __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v2l *src = static_cast<const v2l *>(in) + f * 1024 + lane_off + (q & 1) * ow0 + ((q >> 1) & 1) * ow1 + (q >> 2) * ow2;
                    v2l x0 = {0, 0}, x1 = {0, 0};
                    if (ok) x0 = INTFFT_LD(src), x1 = INTFFT_LD(src + 1);
                    re[q] = wrapw<int64_t>((int64_t)x0.x, a.dw), im[q] = wrapw<int64_t>((int64_t)x0.y, a.dw);
                    re[q + 8] = wrapw<int64_t>((int64_t)x1.x, a.dw), im[q + 8] = wrapw<int64_t>((int64_t)x1.y, a.dw);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                swap32_64(re[r], re[r + 8]);
                swap32_64(im[r], im[r + 8]);
            }
        } else if (a.in_cb == 4) { // L == 10
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + f * 1024 + lane;
            const int sh = 32 - a.dw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                const v2i x = INTFFT_LD(src + 64 * rr);
                re[r] = (int)((u32)x.x << sh) >> sh;
                im[r] = (int)((u32)x.y << sh) >> sh;
            }
        } else {
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            const v2l *src = static_cast<const v2l *>(in) + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                const v2l x = INTFFT_LD(src + 64 * rr);
                re[r] = wrapw<int64_t>((int64_t)x.x, a.dw);
                im[r] = wrapw<int64_t>((int64_t)x.y, a.dw);
            }
        }
        // ---- DIT 0, 1 (multiplier-free), 2, 3 (wave-uniform twiddles) on register offsets 1, 2, 4, 8 ----
#pragma unroll
        for (int g = 0; g < 16; g += 2) fly64<RNDC, 1, CM, true>(a.st[0], 0, re[g], im[g], re[g + 1], im[g + 1], 0, 0);
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            fly64<RNDC, 1, CM, true>(a.st[1], 0, re[g], im[g], re[g + 2], im[g + 2], 0, 0);
            fly64<RNDC, 1, CM, true>(a.st[1], 1, re[g + 1], im[g + 1], re[g + 3], im[g + 3], 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) fly64<RNDC, 2, CM, true>(a.st[2], 0, re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) fly64<RNDC, 2, CM, true>(a.st[3], 0, re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r]);
        // ---- LDS transpose (four dword planes) ----
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            wr_inv[ROWU * r] = (u32)re[r];
            wr_inv[PLANE64 + ROWU * r] = (u32)((unsigned long long)re[r] >> 32);
            wr_inv[2 * PLANE64 + ROWU * r] = (u32)im[r];
            wr_inv[3 * PLANE64 + ROWU * r] = (u32)((unsigned long long)im[r] >> 32);
        }
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd_base[q], xh = rd_base[q + PLANE64 / 4], y = rd_base[q + 2 * (PLANE64 / 4)], yh = rd_base[q + 3 * (PLANE64 / 4)];
            const u32 xl[4] = {x.x, x.y, x.z, x.w}, xu[4] = {xh.x, xh.y, xh.z, xh.w}, yl[4] = {y.x, y.y, y.z, y.w}, yu[4] = {yh.x, yh.y, yh.z, yh.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                re[4 * q + i] = (i64)(((unsigned long long)xu[i] << 32) | xl[i]);
                im[4 * q + i] = (i64)(((unsigned long long)yu[i] << 32) | yl[i]);
            }
        }
        wave_lds_fence();
        // ---- DIT 4 (reg bit 2 = a4), lane bit 4 <-> reg bit 2, DIT 5 (reg bit 3 = a5), lane bit 5 <-> reg bit 3 ----
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) fly64<RNDC, 2, CM, true>(a.st[4], 0, re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r, w4i);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                swap16_64(re[g + j], re[g + j + 4]);
                swap16_64(im[g + j], im[g + j + 4]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) fly64<RNDC, 2, CM, true>(a.st[5], 0, re[j], im[j], re[j + 8], im[j + 8], w5r, w5i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            swap32_64(re[j], re[j + 8]);
            swap32_64(im[j], im[j + 8]);
        }
        // ---- DIT 6..L-1 on register offsets 1, 2, 4, 8 ----
        if constexpr (L >= 7) {
#pragma unroll
            for (int g = 0; g < 16; g += 2) fly64<RNDC, 2, CM, true>(a.st[6], 0, re[g], im[g], re[g + 1], im[g + 1], w6r, w6i);
        }
        if constexpr (L >= 8) {
#pragma unroll
            for (int g = 0; g < 16; g += 4)
#pragma unroll
                for (int j = 0; j < 2; ++j) fly64<RNDC, 2, CM, true>(a.st[7], 0, re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w7r[j], w7i[j]);
        }
        if constexpr (L >= 9) {
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int j = 0; j < 4; ++j) fly64<RNDC, 2, CM, true>(a.st[8], 0, re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w8r[j], w8i[j]);
        }
        if constexpr (L >= 10) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fly64<RNDC, 2, CM, true>(a.st[9], 0, re[j], im[j], re[j + 8], im[j + 8], w9r[j], w9i[j]);
        }
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        v2l *dst = reinterpret_cast<v2l *>(out) + f * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (!partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) {
                const v2l y = {re[j], im[j]};
                __builtin_nontemporal_store(y, dst + 64 * j);
            }
    }
}

template <typename K>
inline void launch_w64_kernel(K kernel, int log2n, const UConsts &c, const W64Args &a, const void *in, void *out, const int2 *tw_all, size_t nframes,
                              hipStream_t stream)
{
    const size_t chunks = (nframes + ((size_t)1 << (10 - log2n)) - 1) >> (10 - log2n); // one wave per 1024 samples
    const size_t need = (chunks + 3) / 4;
    const size_t ldsb = (size_t)4 * 4 * PLANE64 * sizeof(u32);
    allow_max_lds(kptr(kernel));
    const size_t cap = resident_blocks(kptr(kernel), 256, 2, 2, false);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(need < cap ? need : cap)), dim3(256), ldsb, stream, in, static_cast<i64 *>(out), tw_all, c, a, nframes);
}

hipError_t launch_fastw64_short(int log2n, int direction, int rnd_kind, const UConsts &c, const W64Args &a, const void *in, void *out,
                                const int2 *tw_all, size_t nframes, hipStream_t stream);

} // namespace intfft
