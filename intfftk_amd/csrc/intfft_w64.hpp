// intfft_w64.hpp -- the 64-bit kernels as templates: wave kernels (intfft_fastw64.hip: N = 1024 and the launchers; intfft_fastw64s.hip:
// N = 64 .. 512) and, further down, block kernels (intfft_fastw64b.hip / intfft_fastw64bi.hip: N = 2048 / 4096).
// Wave kernels: int_fftNk / int_ifftNk with NFFT = 6 .. 10 (N = 64 .. 1024), natural order in and out, for every DATA_WIDTH /
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
    int native;       // NAT instantiations (N >= 128): bit 0 HALVES order on the time side, bit 1 BITREV order on the frequency side
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

// NAT (round 5, N = 128 .. 1024): the cores' own beat orders.  HALVES in: the register pair (j, j + 2^(L-7)) is the sample pair (n, n + N/2) of one beat, one
// 16- / 32-byte load.  BITREV out: a lane's 16 results are 16 consecutive positions of the chunk (rows of the frames one after the other), so they go through the
// wave's LDS tile (rows of 16 samples, 68 dwords apart) and leave as 1 KiB per store instruction like the natural order.
template <int L, int RNDC, int CM, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W64_WAVES, W64_WAVES))) void k_fft1024_w64(const void *in, i64 *out, const int2 *__restrict__ twt, const UConsts c, const W64Args a,
                                                     size_t nframes_user)
{
    static_assert(!NAT || L >= 7, "native beat orders: N >= 128");
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[]; // 4 waves x 4 planes x 64 rows x ROWU
    const bool halves = NAT && (a.native & 1), bitrev = NAT && (a.native & 2);
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
    int nat_row = 0; // NAT, BITREV side: the lane's positions (frame bits on top) >> 4
#pragma unroll
    for (int k = 4; k < 10; ++k) nat_row |= ((lane >> lane_bit_u<L>(k)) & 1) << (k - 4);

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user; // last chunk: absent frames read as 0
        i64 re[16], im[16];
        if (NAT && halves) {
            constexpr int HB = 1 << (L >= 7 ? L - 7 : 0); // register bit that carries a(L-1)
            const int sh = 32 - a.dw;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB), p0 = 64 * j0, pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                const bool ok = !partial || f * FP + (size_t)(p0 >> L) < nframes_user;
                if (a.in_cb == 4) {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    v4i x = {0, 0, 0, 0};
                    if (ok) x = INTFFT_LD(static_cast<const v4i *>(in) + f * 512 + pair + lane);
                    re[j0] = (int)((u32)x.x << sh) >> sh, im[j0] = (int)((u32)x.y << sh) >> sh;
                    re[j0 + HB] = (int)((u32)x.z << sh) >> sh, im[j0 + HB] = (int)((u32)x.w << sh) >> sh;
                } else {
                    typedef i64 v2l __attribute__((ext_vector_type(2)));
                    const v2l *src = static_cast<const v2l *>(in) + f * 1024 + 2 * (pair + lane);
                    v2l x0 = {0, 0}, x1 = {0, 0};
                    if (ok) x0 = INTFFT_LD(src), x1 = INTFFT_LD(src + 1);
                    re[j0] = wrapw<int64_t>((int64_t)x0.x, a.dw), im[j0] = wrapw<int64_t>((int64_t)x0.y, a.dw);
                    re[j0 + HB] = wrapw<int64_t>((int64_t)x1.x, a.dw), im[j0 + HB] = wrapw<int64_t>((int64_t)x1.y, a.dw);
                }
            }
        } else if (a.in_cb == 4) {
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
        if (NAT && bitrev) { // memory index = position = 16 nat_row + r (N = 1024: nat_row = rev6(lane))
            v2l *row = reinterpret_cast<v2l *>(lds + 68 * nat_row);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const v2l y = {re[r], im[r]};
                row[r] = y;
            }
            wave_lds_fence();
            v2l *dst = reinterpret_cast<v2l *>(out) + f * 1024 + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const v2l y = *reinterpret_cast<const v2l *>(lds + 68 * (4 * i + (lane >> 4)) + 4 * (lane & 15));
                if (!partial || f * FP + (size_t)((64 * i + lane) >> L) < nframes_user) __builtin_nontemporal_store(y, dst + 64 * i);
            }
            wave_lds_fence();
        } else if constexpr (L < 10) { // lane bit 5 <-> reg bit 3: every lane holds pairs of consecutive outputs (32 bytes)
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
// NAT: BITREV in through the wave's LDS tile (the mirror of the forward kernel's store), HALVES out as one 32-byte store per register pair.
template <int L, int RNDC, int CM, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W64_WAVES, W64_WAVES))) void k_ifft1024_w64(const void *in, i64 *out, const int2 *__restrict__ twt, const UConsts c, const W64Args a,
                                                      size_t nframes_user)
{
    static_assert(!NAT || L >= 7, "native beat orders: N >= 128");
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[];
    const bool halves = NAT && (a.native & 1), bitrev = NAT && (a.native & 2);
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
    int nat_row = 0; // NAT, BITREV side: the lane's positions (frame bits on top) >> 4
#pragma unroll
    for (int k = 4; k < 10; ++k) nat_row |= ab(k) << (k - 4);

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user;
        i64 re[16], im[16];
        if (NAT && bitrev) { // memory index = position = 16 nat_row + r (N = 1024: nat_row = rev6(lane))
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            wave_lds_fence();
            if (a.in_cb == 4) {
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i *src = static_cast<const v2i *>(in) + f * 1024 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v2i x = {0, 0};
                    if (!partial || f * FP + (size_t)((64 * i + lane) >> L) < nframes_user) x = INTFFT_LD(src + 64 * i);
                    *reinterpret_cast<v2i *>(lds + 68 * (4 * i + (lane >> 4)) + 4 * (lane & 15)) = x;
                }
            } else {
                const v2l *src = static_cast<const v2l *>(in) + f * 1024 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v2l x = {0, 0};
                    if (!partial || f * FP + (size_t)((64 * i + lane) >> L) < nframes_user) x = INTFFT_LD(src + 64 * i);
                    *reinterpret_cast<v2l *>(lds + 68 * (4 * i + (lane >> 4)) + 4 * (lane & 15)) = x;
                }
            }
            wave_lds_fence();
            const u32 *row = lds + 68 * nat_row;
            if (a.in_cb == 4) {
                typedef int v2i __attribute__((ext_vector_type(2)));
                const int sh = 32 - a.dw;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const v2i x = *reinterpret_cast<const v2i *>(row + 4 * r);
                    re[r] = (int)((u32)x.x << sh) >> sh;
                    im[r] = (int)((u32)x.y << sh) >> sh;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const v2l x = *reinterpret_cast<const v2l *>(row + 4 * r);
                    re[r] = wrapw<int64_t>((int64_t)x.x, a.dw);
                    im[r] = wrapw<int64_t>((int64_t)x.y, a.dw);
                }
            }
        } else if constexpr (L < 10) {
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
        if (NAT && halves) {
            constexpr int HB = 1 << (L >= 7 ? L - 7 : 0);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB), p0 = 64 * j0, pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                if (!partial || f * FP + (size_t)(p0 >> L) < nframes_user) {
                    v2l *dst2 = reinterpret_cast<v2l *>(out) + f * 1024 + 2 * (pair + lane);
                    const v2l y0 = {re[j0], im[j0]}, y1 = {re[j0 + HB], im[j0 + HB]};
                    __builtin_nontemporal_store(y0, dst2);
                    __builtin_nontemporal_store(y1, dst2 + 1);
                }
            }
            continue;
        }
        v2l *dst = reinterpret_cast<v2l *>(out) + f * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (!partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) {
                const v2l y = {re[j], im[j]};
                __builtin_nontemporal_store(y, dst + 64 * j);
            }
    }
}

// ---- block kernels, N = 2048 / 4096 (intfft_fastw64b.hip) ------------------------------------------------------------------------
// Workgroup mapping of intfft_fast4096w.hip / intfft_w32inv.hip: 256 threads own 4096 consecutive samples (one frame, or two 2048-point
// frames whose frame-number stage is skipped), 16 samples per thread as 64-bit register pairs, three register rounds of four stages
//   LA  reg = n11..8, thread = n7..0            STAGE 11..8
//   LB  reg = n7..4,  thread = (n11..8, n3..0)  STAGE 7..4
//   LC  reg = n3..0,  thread = lc64_bit<L>()    STAGE 3..0, the bit reversal folded into the mapping
// with block-wide LDS transposes: one component (re, then im) at a time through two dword planes (lo, hi) of one 40 KiB region --
// four planes at once would be 80 KiB per workgroup, all of a CU's LDS for the two workgroups the 256-VGPR budget allows.
// The 30 thread-dependent twiddle pairs are frame invariant and live in VGPRs.
struct W64BArgs {
    StageDesc st[12]; // indexed by the STAGE generic
    int in_cb;        // input container bytes per component: 4 or 8
    int dw;           // DATA_WIDTH
    int native;       // NAT instantiations: bit 0 HALVES order on the time side, bit 1 BITREV order on the frequency side
};

constexpr int ROW64B = 20;
constexpr int PLANE64B = 256 * ROW64B;
template <int L> __host__ __device__ constexpr int lc64_bit(int k) { return k < L ? (L - 1) - k : (L - 4) + (k - L); }
template <int L> __host__ __device__ constexpr int lc64_row_of_reg(int j)
{
    return (((j >> 0) & 1) << lc64_bit<L>(4)) | (((j >> 1) & 1) << lc64_bit<L>(5)) | (((j >> 2) & 1) << lc64_bit<L>(6)) |
           (((j >> 3) & 1) << lc64_bit<L>(7));
}
__device__ __forceinline__ constexpr int rev4b(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// one register round of the forward core: stages S0 + 3 .. S0 on register offsets 8, 4, 2, 1 (only those below L)
template <int RNDC, int CM, int L, int S0>
__device__ __forceinline__ void round64_dif(i64 (&re)[16], i64 (&im)[16], const int (&w8r)[8], const int (&w8i)[8], const int (&w4r)[4],
                                            const int (&w4i)[4], const int (&w2r)[2], const int (&w2i)[2], int w1r, int w1i, const W64BArgs &a)
{
    if constexpr (S0 + 3 < L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) fly64<RNDC, 2, CM>(a.st[S0 + 3], 0, re[j], im[j], re[j + 8], im[j + 8], w8r[j], w8i[j]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) fly64<RNDC, 2, CM>(a.st[S0 + 2], 0, re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r[j], w4i[j]);
#pragma unroll
    for (int g = 0; g < 16; g += 4)
#pragma unroll
        for (int j = 0; j < 2; ++j) fly64<RNDC, 2, CM>(a.st[S0 + 1], 0, re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w2r[j], w2i[j]);
#pragma unroll
    for (int g = 0; g < 16; g += 2) fly64<RNDC, 2, CM>(a.st[S0], 0, re[g], im[g], re[g + 1], im[g + 1], w1r, w1i);
}
// ... and of the inverse core: stages S0 .. S0 + 3 on register offsets 1, 2, 4, 8
template <int RNDC, int CM, int L, int S0>
__device__ __forceinline__ void round64_dit(i64 (&re)[16], i64 (&im)[16], const int (&w8r)[8], const int (&w8i)[8], const int (&w4r)[4],
                                            const int (&w4i)[4], const int (&w2r)[2], const int (&w2i)[2], int w1r, int w1i, const W64BArgs &a)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2) fly64<RNDC, 2, CM, true>(a.st[S0], 0, re[g], im[g], re[g + 1], im[g + 1], w1r, w1i);
#pragma unroll
    for (int g = 0; g < 16; g += 4)
#pragma unroll
        for (int j = 0; j < 2; ++j) fly64<RNDC, 2, CM, true>(a.st[S0 + 1], 0, re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w2r[j], w2i[j]);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) fly64<RNDC, 2, CM, true>(a.st[S0 + 2], 0, re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r[j], w4i[j]);
    if constexpr (S0 + 3 < L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) fly64<RNDC, 2, CM, true>(a.st[S0 + 3], 0, re[j], im[j], re[j + 8], im[j + 8], w8r[j], w8i[j]);
    }
}

// the thread-dependent twiddles of rounds LA (thread = n7..0: STAGE 11 index 256 jj + tid .. STAGE 8 index tid) and LB (thread low
// nibble = n3..0: STAGE 7 index 16 jj + lo4 .. STAGE 4 index lo4)
struct Tw64B {
    int a8r[8], a8i[8], a4r[4], a4i[4], a2r[2], a2i[2], a1r, a1i;
    int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
};
template <int L> __device__ __forceinline__ void load_tw64_la(Tw64B &t, const int2 *__restrict__ twt, int tid)
{
    int2 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.a8r[j] = t.a8i[j] = 0;
    if constexpr (L >= 12) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w = twt[2047 + 256 * j + tid], t.a8r[j] = w.x, t.a8i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w = twt[1023 + 256 * j + tid], t.a4r[j] = w.x, t.a4i[j] = w.y;
#pragma unroll
    for (int j = 0; j < 2; ++j) w = twt[511 + 256 * j + tid], t.a2r[j] = w.x, t.a2i[j] = w.y;
    w = twt[255 + tid], t.a1r = w.x, t.a1i = w.y;
}
__device__ __forceinline__ void load_tw64_lb(Tw64B &t, const int2 *__restrict__ twt, int lo4)
{
    int2 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) w = twt[127 + 16 * j + lo4], t.b8r[j] = w.x, t.b8i[j] = w.y;
#pragma unroll
    for (int j = 0; j < 4; ++j) w = twt[63 + 16 * j + lo4], t.b4r[j] = w.x, t.b4i[j] = w.y;
#pragma unroll
    for (int j = 0; j < 2; ++j) w = twt[31 + 16 * j + lo4], t.b2r[j] = w.x, t.b2i[j] = w.y;
    w = twt[15 + lo4], t.b1r = w.x, t.b1i = w.y;
}
// round LA's 15 pairs stay in registers over the frame loop where the butterflies leave room (narrow products, truncate / unscaled);
// the three-dword and round-mode bodies re-read them per frame from the L2-resident table instead (they spill 50-320 dwords otherwise)
template <int RNDC, int CM> constexpr bool keep_la64() { return CM == 1 && RNDC != RND_ROUND; }
template <int RNDC, int CM> constexpr bool keep_lb64() { return CM == 1; } // (three-dword products: round LB's 15 pairs too)

// block-wide transpose of one 64-bit component: register j of every thread goes to wbase[woff(j)] (lo plane; hi plane PLANE64B above),
// thread t then reads row t (16 consecutive values)
template <typename OFF>
__device__ __forceinline__ void xpose64(i64 (&v)[16], u32 *lds, u32 *wbase, OFF woff, int tid)
{
    __syncthreads(); // the previous reads of the region are done
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        wbase[woff(j)] = (u32)v[j];
        wbase[PLANE64B + woff(j)] = (u32)((unsigned long long)v[j] >> 32);
    }
    __syncthreads();
    const uint4 *rd = reinterpret_cast<const uint4 *>(lds + ROW64B * tid);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 x = rd[q], h = rd[q + PLANE64B / 4];
        const u32 xl[4] = {x.x, x.y, x.z, x.w}, xh[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * q + i] = (i64)(((unsigned long long)xh[i] << 32) | xl[i]);
    }
}

// NAT (round 5): the cores' own beat orders.  HALVES on the time side: the register pair (j, j + 2^(L-9)) of layout LA is the sample pair (n, n + N/2) of one
// beat, one 16- / 32-byte access.  BITREV on the frequency side: memory index = core position, an LC thread owns the 16 consecutive positions of row
// `rowp` -- one component at a time through the transpose region (rows of 16 int64 + 2 dwords, slots XOR-swizzled by the row's two high bits so that the
// lanes of a wave spread over the banks), so that every global instruction of the workgroup moves 4 KiB like the natural order.
constexpr int ROWNB64 = 34;
__device__ __forceinline__ int nb64_slot(int row, int slot) { return ROWNB64 * row + 2 * (slot ^ ((row >> 6) & 3)); }
__device__ __forceinline__ void rows_to_linear64(i64 (&v)[16], u32 *lds, int rowp, int tid) // v[r] = position 16 rowp + r  ->  v[i] = position 256 i + tid
{
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) *reinterpret_cast<i64 *>(lds + nb64_slot(rowp, r)) = v[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const i64 *>(lds + nb64_slot(16 * i + (tid >> 4), tid & 15));
}
__device__ __forceinline__ void linear_to_rows64(i64 (&v)[16], u32 *lds, int rowp, int tid) // ... and back
{
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<i64 *>(lds + nb64_slot(16 * i + (tid >> 4), tid & 15)) = v[i];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = *reinterpret_cast<const i64 *>(lds + nb64_slot(rowp, r));
}

template <int L, int RNDC, int CM, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fft4096_w64(const void *in, i64 *out, const int2 *__restrict__ twt,
                                                                                                 const UConsts c, const W64BArgs a, size_t nframes_user)
{
    static_assert(L == 11 || L == 12, "block kernel: N = 2048 or 4096");
    constexpr int FP = 1 << (12 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks of 4096 samples
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANE64B];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    const bool halves = NAT && (a.native & 1), bitrev = NAT && (a.native & 2);
    Tw64B t;
    if constexpr (keep_lb64<RNDC, CM>()) load_tw64_lb(t, twt, lo4);
    if constexpr (keep_la64<RNDC, CM>()) load_tw64_la<L>(t, twt, tid);
    // LA -> LB: element (thread x, reg y) -> row 16 y + x3..0, column x7..4
    u32 *const w_ab = lds + ROW64B * lo4 + hi4;
    // LB -> LC: thread (hi4 = n11..8, lo4 = n3..0), reg j' = n7..4 -> row = LC thread (lc64_bit<L>), column n3..0
    const int row_hi = ((hi4 & 1) << lc64_bit<L>(8)) | (((hi4 >> 1) & 1) << lc64_bit<L>(9)) | (((hi4 >> 2) & 1) << lc64_bit<L>(10)) |
                       (((hi4 >> 3) & 1) << lc64_bit<L>(11));
    u32 *const w_bc = lds + ROW64B * row_hi + lo4;
    int lc_off = 0, lc_frame = 0, rowp = 0; // LC <-> natural-order X: index = rev4(r) * 2^(L-4) + lc_off; rowp: the thread's core positions >> 4
#pragma unroll
    for (int k = 4; k < 12; ++k) {
        const int bit = (tid >> lc64_bit<L>(k)) & 1;
        lc_off += bit * (k >= L ? (1 << k) : (1 << (L - 1 - k)));
        if (k >= L) lc_frame += bit << (k - L);
        rowp += bit << (k - 4);
    }
    auto off_ab = [](int j) { return ROW64B * 16 * j; };
    auto off_bc = [](int j) { return ROW64B * lc64_row_of_reg<L>(j); };

    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const bool partial = L < 12 && (f + 1) * FP > nframes_user; // last chunk: the absent frame reads as 0
        i64 re[16], im[16];
        if (NAT && halves) {
            constexpr int HB = 1 << (L - 9);
            const int sh = 32 - a.dw;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB), p0 = 256 * j0, pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                const bool ok = !partial || f * FP + (size_t)(p0 >> L) < nframes_user;
                if (a.in_cb == 4) {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    v4i x = {0, 0, 0, 0};
                    if (ok) x = INTFFT_LD(static_cast<const v4i *>(in) + f * 2048 + pair + tid);
                    re[j0] = (int)((u32)x.x << sh) >> sh, im[j0] = (int)((u32)x.y << sh) >> sh;
                    re[j0 + HB] = (int)((u32)x.z << sh) >> sh, im[j0 + HB] = (int)((u32)x.w << sh) >> sh;
                } else {
                    typedef i64 v2l __attribute__((ext_vector_type(2)));
                    const v2l *src = static_cast<const v2l *>(in) + f * 4096 + 2 * (pair + tid);
                    v2l x0 = {0, 0}, x1 = {0, 0};
                    if (ok) x0 = INTFFT_LD(src), x1 = INTFFT_LD(src + 1);
                    re[j0] = wrapw<int64_t>((int64_t)x0.x, a.dw), im[j0] = wrapw<int64_t>((int64_t)x0.y, a.dw);
                    re[j0 + HB] = wrapw<int64_t>((int64_t)x1.x, a.dw), im[j0 + HB] = wrapw<int64_t>((int64_t)x1.y, a.dw);
                }
            }
        } else if (a.in_cb == 4) {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + f * 4096 + tid;
            const int sh = 32 - a.dw;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v2i x = {0, 0};
                if (!partial || f * FP + (size_t)((256 * j + tid) >> L) < nframes_user) x = INTFFT_LD(src + 256 * j);
                re[j] = (int)((u32)x.x << sh) >> sh;
                im[j] = (int)((u32)x.y << sh) >> sh;
            }
        } else {
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            const v2l *src = static_cast<const v2l *>(in) + f * 4096 + tid;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v2l x = {0, 0};
                if (!partial || f * FP + (size_t)((256 * j + tid) >> L) < nframes_user) x = INTFFT_LD(src + 256 * j);
                re[j] = wrapw<int64_t>((int64_t)x.x, a.dw);
                im[j] = wrapw<int64_t>((int64_t)x.y, a.dw);
            }
        }
        if constexpr (!keep_la64<RNDC, CM>()) {
            const int2 *tp = twt;
            asm volatile("" : "+s"(tp)); // (opaque: the loads stay inside the frame loop)
            load_tw64_la<L>(t, tp, tid);
        }
        round64_dif<RNDC, CM, L, 8>(re, im, t.a8r, t.a8i, t.a4r, t.a4i, t.a2r, t.a2i, t.a1r, t.a1i, a);
        xpose64(re, lds, w_ab, off_ab, tid);
        xpose64(im, lds, w_ab, off_ab, tid);
        if constexpr (!keep_lb64<RNDC, CM>()) {
            const int2 *tp = twt;
            asm volatile("" : "+s"(tp));
            load_tw64_lb(t, tp, lo4);
        }
        round64_dif<RNDC, CM, 12, 4>(re, im, t.b8r, t.b8i, t.b4r, t.b4i, t.b2r, t.b2i, t.b1r, t.b1i, a);
        xpose64(re, lds, w_bc, off_bc, tid);
        xpose64(im, lds, w_bc, off_bc, tid);
        // LC: stages 3, 2 (wave-uniform twiddles), 1, 0 (multiplier-free)
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
        if (NAT && bitrev) {
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            rows_to_linear64(re, lds, rowp, tid);
            rows_to_linear64(im, lds, rowp, tid);
            v2l *dst = reinterpret_cast<v2l *>(out) + f * 4096 + tid;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!partial || f * FP + (size_t)((256 * i + tid) >> L) < nframes_user) {
                    const v2l y = {re[i], im[i]};
                    __builtin_nontemporal_store(y, dst + 256 * i);
                }
        } else if (!partial || f * FP + (size_t)lc_frame < nframes_user) {
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            v2l *dst = reinterpret_cast<v2l *>(out) + f * 4096 + lc_off;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const v2l y = {re[r], im[r]};
                __builtin_nontemporal_store(y, dst + (rev4b(r) << (L - 4)));
            }
        }
    }
}

template <int L, int RNDC, int CM, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ifft4096_w64(const void *in, i64 *out, const int2 *__restrict__ twt,
                                                                                                  const UConsts c, const W64BArgs a, size_t nframes_user)
{
    static_assert(L == 11 || L == 12, "block kernel: N = 2048 or 4096");
    constexpr int FP = 1 << (12 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP;
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANE64B];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    const bool halves = NAT && (a.native & 1), bitrev = NAT && (a.native & 2);
    Tw64B t;
    if constexpr (keep_lb64<RNDC, CM>()) load_tw64_lb(t, twt, lo4);
    if constexpr (keep_la64<RNDC, CM>()) load_tw64_la<L>(t, twt, tid);
    // LC thread carries n_k on bit lc64_bit<L>(k); LC -> LB: row = LB thread 16 (n11..8) + r, column = n7..4
    auto nb = [&](int k) { return (tid >> lc64_bit<L>(k)) & 1; };
    const int lb_hi = nb(8) | (nb(9) << 1) | (nb(10) << 2) | (nb(11) << 3), lb_reg = nb(4) | (nb(5) << 1) | (nb(6) << 2) | (nb(7) << 3);
    u32 *const w_cb = lds + ROW64B * 16 * lb_hi + lb_reg;
    // LB -> LA: element (thread (n11..8 = hi4, n3..0 = lo4), reg n7..4) -> row n7..0 = 16 j' + lo4, column n11..8 = hi4
    u32 *const w_ba = lds + ROW64B * lo4 + hi4;
    int lc_off = 0, lc_frame = 0, rowp = 0;
#pragma unroll
    for (int k = 4; k < 12; ++k) {
        lc_off += nb(k) * (k >= L ? (1 << k) : (1 << (L - 1 - k)));
        if (k >= L) lc_frame += nb(k) << (k - L);
        rowp += nb(k) << (k - 4);
    }
    auto off_cb = [](int r) { return ROW64B * r; };
    auto off_ba = [](int j) { return ROW64B * 16 * j; };

    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const bool partial = L < 12 && (f + 1) * FP > nframes_user;
        const bool lc_ok = !partial || f * FP + (size_t)lc_frame < nframes_user;
        i64 re[16], im[16];
        // LC: X[brev_L(n)] of the thread's frame
        if (NAT && bitrev) { // memory index = core position: coalesced loads, one component at a time through the transpose region
            if (a.in_cb == 4) {
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i *src = static_cast<const v2i *>(in) + f * 4096 + tid;
                const int sh = 32 - a.dw;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v2i x = {0, 0};
                    if (!partial || f * FP + (size_t)((256 * i + tid) >> L) < nframes_user) x = INTFFT_LD(src + 256 * i);
                    re[i] = (int)((u32)x.x << sh) >> sh;
                    im[i] = (int)((u32)x.y << sh) >> sh;
                }
            } else {
                typedef i64 v2l __attribute__((ext_vector_type(2)));
                const v2l *src = static_cast<const v2l *>(in) + f * 4096 + tid;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v2l x = {0, 0};
                    if (!partial || f * FP + (size_t)((256 * i + tid) >> L) < nframes_user) x = INTFFT_LD(src + 256 * i);
                    re[i] = wrapw<int64_t>((int64_t)x.x, a.dw);
                    im[i] = wrapw<int64_t>((int64_t)x.y, a.dw);
                }
            }
            linear_to_rows64(re, lds, rowp, tid);
            linear_to_rows64(im, lds, rowp, tid);
        } else if (a.in_cb == 4) {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + f * 4096 + lc_off;
            const int sh = 32 - a.dw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v2i x = {0, 0};
                if (lc_ok) x = INTFFT_LD(src + (rev4b(r) << (L - 4)));
                re[r] = (int)((u32)x.x << sh) >> sh;
                im[r] = (int)((u32)x.y << sh) >> sh;
            }
        } else {
            typedef i64 v2l __attribute__((ext_vector_type(2)));
            const v2l *src = static_cast<const v2l *>(in) + f * 4096 + lc_off;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v2l x = {0, 0};
                if (lc_ok) x = INTFFT_LD(src + (rev4b(r) << (L - 4)));
                re[r] = wrapw<int64_t>((int64_t)x.x, a.dw);
                im[r] = wrapw<int64_t>((int64_t)x.y, a.dw);
            }
        }
        // DIT 0, 1 (multiplier-free), 2, 3 (wave-uniform twiddles) on register offsets 1, 2, 4, 8
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
        xpose64(re, lds, w_cb, off_cb, tid); // LB: regs = n7..4
        xpose64(im, lds, w_cb, off_cb, tid);
        if constexpr (!keep_lb64<RNDC, CM>()) {
            const int2 *tp = twt;
            asm volatile("" : "+s"(tp));
            load_tw64_lb(t, tp, lo4);
        }
        round64_dit<RNDC, CM, 12, 4>(re, im, t.b8r, t.b8i, t.b4r, t.b4i, t.b2r, t.b2i, t.b1r, t.b1i, a);
        xpose64(re, lds, w_ba, off_ba, tid); // LA: regs = n11..8, thread = n7..0
        xpose64(im, lds, w_ba, off_ba, tid);
        if constexpr (!keep_la64<RNDC, CM>()) {
            const int2 *tp = twt;
            asm volatile("" : "+s"(tp));
            load_tw64_la<L>(t, tp, tid);
        }
        round64_dit<RNDC, CM, L, 8>(re, im, t.a8r, t.a8i, t.a4r, t.a4i, t.a2r, t.a2i, t.a1r, t.a1i, a);
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        if (NAT && halves) {
            constexpr int HB = 1 << (L - 9);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB), p0 = 256 * j0, pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                if (!partial || f * FP + (size_t)(p0 >> L) < nframes_user) {
                    v2l *dst2 = reinterpret_cast<v2l *>(out) + f * 4096 + 2 * (pair + tid);
                    const v2l y0 = {re[j0], im[j0]}, y1 = {re[j0 + HB], im[j0 + HB]};
                    __builtin_nontemporal_store(y0, dst2);
                    __builtin_nontemporal_store(y1, dst2 + 1);
                }
            }
            continue;
        }
        v2l *dst = reinterpret_cast<v2l *>(out) + f * 4096 + tid;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (!partial || f * FP + (size_t)((256 * j + tid) >> L) < nframes_user) {
                const v2l y = {re[j], im[j]};
                __builtin_nontemporal_store(y, dst + 256 * j);
            }
    }
}

template <typename K>
inline void launch_w64b_kernel(K kernel, int log2n, const UConsts &c, const W64BArgs &a, const void *in, void *out, const int2 *tw_all, size_t nframes,
                               hipStream_t stream)
{
    const size_t chunks = (nframes + ((size_t)1 << (12 - log2n)) - 1) >> (12 - log2n); // one workgroup per 4096 samples
    const size_t cap = resident_blocks(kptr(kernel), 256, 2, 2, false);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(chunks < cap ? chunks : cap)), dim3(256), 0, stream, in, static_cast<i64 *>(out), tw_all, c, a, nframes);
}

hipError_t launch_fastw64_block(int log2n, int direction, int rnd_kind, int cm, const UConsts &c, const W64BArgs &a, const void *in, void *out,
                                const int2 *tw_all, size_t nframes, hipStream_t stream);
hipError_t launch_fastw64_block_inv(int log2n, int rnd_kind, int cm, const UConsts &c, const W64BArgs &a, const void *in, void *out,
                                    const int2 *tw_all, size_t nframes, hipStream_t stream);
hipError_t launch_fastw64_block_native(int log2n, int direction, int rnd_kind, int cm, const UConsts &c, const W64BArgs &a, const void *in, void *out,
                                       const int2 *tw_all, size_t nframes, hipStream_t stream); // intfft_fastw64bn.hip

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
hipError_t launch_fastw64_short_native(int log2n, int direction, int rnd_kind, const UConsts &c, const W64Args &a, const void *in, void *out,
                                       const int2 *tw_all, size_t nframes, hipStream_t stream); // intfft_fastw64sn.hip
hipError_t launch_fastw64_native(int direction, int rnd_kind, int cm, const UConsts &c, const W64Args &a, const void *in, void *out, const int2 *tw_all,
                                 size_t nframes, hipStream_t stream); // intfft_fastw64n.hip

} // namespace intfft
