// intfft_fastw32.hip -- general-width wave kernel: int_fftNk for 64 <= N <= 1024 with ANY DATA_WIDTH / TWDL_WIDTH /
// FORMAT / RNDMODE whose widths stay within 32 bits (DATA_WIDTH + FORMAT * NFFT <= 32, TWDL_WIDTH <= 26), natural
// order in and out -- e.g. 12- or 14-bit converters, 18/24/32-bit scaled data, 24-bit unscaled short frames.
// (DATA_WIDTH = 16 with TWDL_WIDTH <= 16 has the tuned kernels intfft_fast1024.hip / intfft_fast1024u.hip.)
//
// Same wave mapping as intfft_fast1024u.hip (regs a9..6, two lane swaps, one wave-private LDS transpose of two dword
// planes, regs a3..0; short frames share a wave) on unpacked int32 registers, with the per-stage arithmetic fully
// parameterised (wave-uniform W32Stage, from the planner's StageDesc):
//   sum/difference   trunc (A>>1) +/- (B>>1) | round rhu2(A +/- B) wrapped to DTW bits | unscaled A +/- B
//                    (int_dif2_fly.vhd:144-241)
//   multiplier       every regime of int_cmult_dsp48.vhd:182-434 in the masked form of intfft_wide16.hip:
//                    ((M2 & K) -/+ (M1 & K)) >> (a + b), K = ~(2^a - 1); exact 64-bit sums from v_mad_i64_i32
//   width wrap       (x << (32 - w)) >> (32 - w), a no-op when w = 32
// Containers follow the C-ABI: int16 pairs for widths <= 16, int32 pairs above (runtime flags, load/store only).
#include "intfft_u32.hpp"

namespace intfft {

// OUT64 (L = 10, unscaled): 33 / 34-bit results, stages 1, 0 in 64 bits -- a template parameter, so that the 32-bit unscaled
// kernel does not carry the 64-bit tail's registers (192 VGPRs with the runtime branch, two waves per SIMD)
// NAT (round 4): the instantiation for int_fftNk's own beat orders (`native` bit 0: HALVES in, bit 1: BITREV out; N >= 128, results within 32 bits),
// as in intfft_fast1024u.hip: HALVES beats = one 8- / 16-byte load of the register pair (j0, j0 | 2^(L-7)); BITREV order = the core position, 16
// consecutive ones per LC lane, through the wave's LDS tile in memory order (padded rows of 16 samples), 1 KiB per wave instruction
template <int L, int MODE, bool MASKED, bool OUT64 = false, bool NAT = false>
__global__ __launch_bounds__(256) void k_fft1024_w32(const void *in, void *out, const int2 *__restrict__ twt, const UConsts c,
                                                     const W32Args a, size_t nframes_user, int native)
{
    static_assert(!NAT || L >= 7, "native beat orders: N >= 128");
    const bool halves = NAT && (native & 1), bitrev = NAT && (native & 2);
    constexpr int FP = 1 << (10 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks of 1024 samples
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * 2 * 64 * ROWU];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *lds = lds_all + wv * 2 * 64 * ROWU;

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
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    u32 *wr_base = lds + ROWU * ((t5 << lane_bit_u<L>(9)) + (t4 << lane_bit_u<L>(8))) + (lane & 15);
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROWU * lane);
    int lane_off = 0, lane_frame = 0; // N < 1024: see intfft_fast1024u.hip (one output swap level)
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

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user; // last chunk: absent frames read as 0
        int re[16], im[16];
        if (NAT && halves) {
            constexpr int HB = 1 << (L >= 7 ? L - 7 : 0); // register bit that carries a(L-1)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB);
                const int p0 = 64 * j0; // logical position P = p0 + lane (bit L-1 clear): frame P >> L, beat P mod N/2
                const int pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                const bool ok = !partial || f * FP + (size_t)(p0 >> L) < nframes_user;
                if (a.in16) {
                    typedef u32 v2u __attribute__((ext_vector_type(2)));
                    v2u w = {0u, 0u};
                    if (ok) w = INTFFT_LD(reinterpret_cast<const v2u *>(static_cast<const u32 *>(in) + f * 1024) + lane + pair);
                    re[j0] = (int)(w.x << a.in_sh) >> a.in_sh, im[j0] = (int)(w.x << (a.in_sh - 16)) >> a.in_sh;
                    re[j0 | HB] = (int)(w.y << a.in_sh) >> a.in_sh, im[j0 | HB] = (int)(w.y << (a.in_sh - 16)) >> a.in_sh;
                } else {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    v4i w = {0, 0, 0, 0};
                    if (ok) w = INTFFT_LD(reinterpret_cast<const v4i *>(static_cast<const int2 *>(in) + f * 1024) + lane + pair);
                    re[j0] = (int)((u32)w.x << a.in_sh) >> a.in_sh, im[j0] = (int)((u32)w.y << a.in_sh) >> a.in_sh;
                    re[j0 | HB] = (int)((u32)w.z << a.in_sh) >> a.in_sh, im[j0 | HB] = (int)((u32)w.w << a.in_sh) >> a.in_sh;
                }
            }
        } else if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + f * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = !partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user;
                const u32 raw = ok ? INTFFT_LD(src + 64 * j) : 0u;
                re[j] = (int)(raw << a.in_sh) >> a.in_sh; // wrap to DATA_WIDTH (conv_std_logic_vector)
                im[j] = (int)(raw << (a.in_sh - 16)) >> a.in_sh;
            }
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + f * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = !partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user;
                v2i x = {0, 0};
                if (ok) x = INTFFT_LD(src + 64 * j);
                re[j] = (int)((u32)x.x << a.in_sh) >> a.in_sh;
                im[j] = (int)((u32)x.y << a.in_sh) >> a.in_sh;
            }
        }
        // ---- stages 9..6 in registers, 5 and 4 after the lane swaps ----
        if constexpr (L >= 10) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gfly<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w9r[j], w9i[j], a.st[9]);
        }
        if constexpr (L >= 9) {
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int j = 0; j < 4; ++j) gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w8r[j], w8i[j], a.st[8]);
        }
        if constexpr (L >= 8) {
#pragma unroll
            for (int g = 0; g < 16; g += 4)
#pragma unroll
                for (int j = 0; j < 2; ++j) gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w7r[j], w7i[j], a.st[7]);
        }
        if constexpr (L >= 7) {
#pragma unroll
            for (int g = 0; g < 16; g += 2) gfly<MODE, false, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w6r, w6i, a.st[6]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uswap32(re[j], re[j + 8]);
            uswap32(im[j], im[j + 8]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) gfly<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w5r, w5i, a.st[5]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uswap16(re[g + j], re[g + j + 4]);
                uswap16(im[g + j], im[g + j + 4]);
            }
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r, w4i, a.st[4]);
        // ---- LDS transpose: regs become a3..0 ----
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;
            const int row_j = (j1 << lane_bit_u<L>(7)) + (j0 << lane_bit_u<L>(6)) + (j3 << lane_bit_u<L>(5)) + (j2 << lane_bit_u<L>(4));
            wr_base[ROWU * row_j] = (u32)re[j];
            wr_base[64 * ROWU + ROWU * row_j] = (u32)im[j];
        }
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd_base[q], y = rd_base[q + 16 * ROWU];
            re[4 * q + 0] = (int)x.x, re[4 * q + 1] = (int)x.y, re[4 * q + 2] = (int)x.z, re[4 * q + 3] = (int)x.w;
            im[4 * q + 0] = (int)y.x, im[4 * q + 1] = (int)y.y, im[4 * q + 2] = (int)y.z, im[4 * q + 3] = (int)y.w;
        }
        wave_lds_fence();
        // ---- stages 3, 2 (uniform twiddles), 1, 0 ----
#pragma unroll
        for (int r = 0; r < 8; ++r) gfly<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) gfly<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
        if constexpr (L == 10 && MODE == W_UNSCALED && OUT64) { // 33 / 34-bit results: stages 1, 0 in 64 bits, int64 containers
            long long xr[16], xi[16];
            tail64_unscaled(re, im, xr, xi);
            typedef long long v2l __attribute__((ext_vector_type(2)));
            if (NAT && bitrev) {
                // 16 B per sample: the 16 KiB chunk goes through the 10 KiB tile in two halves (a9 = 0, then a9 = 1: a lane's 16 consecutive positions
                // share a9); rows of 16 samples = 64 dwords, 68 apart
                int A = 0;
#pragma unroll
                for (int k = 4; k < 10; ++k) A |= ((lane >> lane_bit_u<L>(k)) & 1) << k;
                v2l *dst2 = reinterpret_cast<v2l *>(out) + f * 1024;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    wave_lds_fence();
                    if ((A >> 9) == h) {
                        v2l *row = reinterpret_cast<v2l *>(lds + 68 * ((A >> 4) & 31));
#pragma unroll
                        for (int r = 0; r < 16; ++r) row[r] = v2l{xr[r], xi[r]};
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = 64 * i + lane; // sample e of this half
                        __builtin_nontemporal_store(*reinterpret_cast<const v2l *>(lds + 68 * (e >> 4) + 4 * (e & 15)), dst2 + 512 * h + e);
                    }
                }
                wave_lds_fence();
                continue;
            }
            v2l *dst = reinterpret_cast<v2l *>(out) + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                const v2l y = {xr[r], xi[r]};
                __builtin_nontemporal_store(y, dst + 64 * rr);
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            gfly_triv<MODE, false>(re[g], im[g], re[g + 2], im[g + 2], a.st[1]);
            gfly_triv<MODE, true>(re[g + 1], im[g + 1], re[g + 3], im[g + 3], a.st[1]);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) gfly_triv<MODE, false>(re[g], im[g], re[g + 1], im[g + 1], a.st[0]);

        // ---- store ----
        if (NAT && bitrev) {
            // position of (lane, reg r) = A(lane) | r: 16 consecutive samples per lane -> one padded row of the wave's LDS tile
            typedef int v4i __attribute__((ext_vector_type(4)));
            int A = 0;
#pragma unroll
            for (int k = 4; k < 10; ++k) A |= ((lane >> lane_bit_u<L>(k)) & 1) << k;
            wave_lds_fence();
            if (a.out16) {
                typedef u32 v4u __attribute__((ext_vector_type(4)));
                u32 *row = lds + 20 * (A >> 4); // 16 samples = 16 dwords per row, 4 dwords of pad
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4u y;
                    y.x = ((u32)re[4 * q] & 0xFFFFu) | ((u32)im[4 * q] << 16), y.y = ((u32)re[4 * q + 1] & 0xFFFFu) | ((u32)im[4 * q + 1] << 16);
                    y.z = ((u32)re[4 * q + 2] & 0xFFFFu) | ((u32)im[4 * q + 2] << 16), y.w = ((u32)re[4 * q + 3] & 0xFFFFu) | ((u32)im[4 * q + 3] << 16);
                    *reinterpret_cast<v4u *>(row + 4 * q) = y;
                }
                wave_lds_fence();
                v4i *dst4 = static_cast<v4i *>(out) + f * 256;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 64 * i + lane; // 16-byte piece = samples 4 e .. 4 e + 3
                    if (L < 10 && f * FP + (size_t)((4 * e) >> L) >= nframes_user) continue;
                    __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 20 * (e >> 2) + 4 * (e & 3)), dst4 + e);
                }
            } else {
                u32 *row = lds + 36 * (A >> 4); // 16 samples = 32 dwords per row, 4 dwords of pad
#pragma unroll
                for (int q = 0; q < 8; ++q) *reinterpret_cast<v4i *>(row + 4 * q) = v4i{re[2 * q], im[2 * q], re[2 * q + 1], im[2 * q + 1]};
                wave_lds_fence();
                v4i *dst4 = static_cast<v4i *>(out) + f * 512;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 64 * i + lane; // 16-byte piece = samples 2 e, 2 e + 1
                    if (L < 10 && f * FP + (size_t)((2 * e) >> L) >= nframes_user) continue;
                    __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 36 * (e >> 3) + 4 * (e & 7)), dst4 + e);
                }
            }
            wave_lds_fence();
        } else if constexpr (L < 10) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uswap32(re[r], re[r + 8]);
                uswap32(im[r], im[r + 8]);
            }
            if constexpr (L == 6) {
                // N = 64: the finished chunk goes through the wave's idle LDS tile in memory order and leaves as 1 KiB per instruction
                // (a lane's pairs lie 32 samples apart: thirty-two short runs per store instruction otherwise; intfft_fast1024u.hip)
                typedef int v4i __attribute__((ext_vector_type(4)));
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int off = lane_off + (q & 1) * out_weight<L>(0) + ((q >> 1) & 1) * out_weight<L>(1) + (q >> 2) * out_weight<L>(2);
                    if (a.out16) *reinterpret_cast<v2u *>(lds + off) = v2u{((u32)re[q] & 0xFFFFu) | ((u32)im[q] << 16), ((u32)re[q + 8] & 0xFFFFu) | ((u32)im[q + 8] << 16)};
                    else *reinterpret_cast<v4i *>(lds + 2 * off) = v4i{re[q], im[q], re[q + 8], im[q + 8]};
                }
                wave_lds_fence();
                const int pieces = a.out16 ? 4 : 8, shift = a.out16 ? L - 2 : L - 1; // 16-byte pieces of the chunk per lane; piece -> frame
                v4i *dst4 = static_cast<v4i *>(out) + f * (a.out16 ? 256 : 512);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i >= pieces) break;
                    const int e = 64 * i + lane;
                    if (f * FP + (size_t)(e >> shift) >= nframes_user) continue;
                    __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 4 * e), dst4 + e);
                }
                wave_lds_fence();
            } else if (f * FP + (size_t)lane_frame < nframes_user) {
                if (a.out16) {
                    typedef u32 v2u __attribute__((ext_vector_type(2)));
                    u32 *dst = static_cast<u32 *>(out) + f * 1024 + lane_off;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const v2u y = {((u32)re[q] & 0xFFFFu) | ((u32)im[q] << 16), ((u32)re[q + 8] & 0xFFFFu) | ((u32)im[q + 8] << 16)};
                        __builtin_nontemporal_store(y, reinterpret_cast<v2u *>(dst + (q & 1) * out_weight<L>(0) +
                                                                               ((q >> 1) & 1) * out_weight<L>(1) +
                                                                               (q >> 2) * out_weight<L>(2)));
                    }
                } else {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    int2 *dst = static_cast<int2 *>(out) + f * 1024 + lane_off;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const v4i y = {re[q], im[q], re[q + 8], im[q + 8]};
                        __builtin_nontemporal_store(y, reinterpret_cast<v4i *>(dst + (q & 1) * out_weight<L>(0) +
                                                                               ((q >> 1) & 1) * out_weight<L>(1) +
                                                                               (q >> 2) * out_weight<L>(2)));
                    }
                }
            }
        } else if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                __builtin_nontemporal_store(((u32)re[r] & 0xFFFFu) | ((u32)im[r] << 16), dst + 64 * rr);
            }
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            int2 *dst = static_cast<int2 *>(out) + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                const v2i y = {re[r], im[r]};
                __builtin_nontemporal_store(y, reinterpret_cast<v2i *>(dst + 64 * rr));
            }
        }
    }
}

bool fastw32_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                       int out_order)
{
    // N = 1024 unscaled: also 33 / 34-bit results (e.g. DATA_WIDTH = 24): only the two multiplier-free stages exceed 32 bits
    const int out_bits = data_width + format * log2n;
    const bool fits = out_bits <= 32 || (log2n == 10 && format == 1 && out_bits <= 34);
    if (!(log2n >= 6 && log2n <= 10 && data_width >= 2 && fits && twdl_width >= 4 && twdl_width <= 26 && direction == 0 && use_fly == 1)) return false;
    if (in_order == 0 && out_order == 0) return true;
    // int_fftNk's own beat orders (HALVES in, BITREV out) and the mixed forms: N >= 128
    return log2n >= 7 && (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1);
}

const char *fastw32_kernel_name() { return "k_fft1024_w32"; }

template <int L, int MODE, bool MASKED, bool OUT64, bool NAT>
static hipError_t launchw_n(const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a, size_t nframes,
                            hipStream_t stream, int native)
{
    const size_t cap = resident_blocks(kptr(k_fft1024_w32<L, MODE, MASKED, OUT64, NAT>), 256, 2);
    const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L);
    const size_t need = (chunks + 3) / 4;
    hipLaunchKernelGGL((k_fft1024_w32<L, MODE, MASKED, OUT64, NAT>), dim3((unsigned)(need < cap ? need : cap)), dim3(256), 0, stream, in, out, tw,
                       c, a, nframes, native);
    return hipGetLastError();
}
template <int L, int MODE, bool MASKED, bool OUT64 = false>
static hipError_t launchw(const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a, size_t nframes,
                          hipStream_t stream, int native)
{
    if constexpr (L >= 7) {
        if (native) return launchw_n<L, MODE, MASKED, OUT64, true>(in, out, tw, c, a, nframes, stream, native);
    }
    return launchw_n<L, MODE, MASKED, OUT64, false>(in, out, tw, c, a, nframes, stream, 0);
}

template <int L>
static hipError_t launchw_l(int mode, const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a,
                            size_t nframes, hipStream_t stream, int native)
{
    if (a.masked) {
        switch (mode) {
        case W_TRUNC: return launchw<L, W_TRUNC, true>(in, out, tw, c, a, nframes, stream, native);
        case W_ROUND: return launchw<L, W_ROUND, true>(in, out, tw, c, a, nframes, stream, native);
        default:
            if constexpr (L == 10)
                if (a.out64) return launchw<L, W_UNSCALED, true, true>(in, out, tw, c, a, nframes, stream, native);
            return launchw<L, W_UNSCALED, true>(in, out, tw, c, a, nframes, stream, native);
        }
    }
    switch (mode) {
    case W_TRUNC: return launchw<L, W_TRUNC, false>(in, out, tw, c, a, nframes, stream, native);
    case W_ROUND: return launchw<L, W_ROUND, false>(in, out, tw, c, a, nframes, stream, native);
    default:
        if constexpr (L == 10)
            if (a.out64) return launchw<L, W_UNSCALED, false, true>(in, out, tw, c, a, nframes, stream, native);
        return launchw<L, W_UNSCALED, false>(in, out, tw, c, a, nframes, stream, native);
    }
}

hipError_t launch_fastw32(int log2n, int mode, const W32Args &a, const void *in, void *out, const int2 *tw_all,
                          const int2 *h_tw, size_t nframes, hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    switch (log2n) {
    case 6: return launchw_l<6>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 7: return launchw_l<7>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 8: return launchw_l<8>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 9: return launchw_l<9>(mode, in, out, tw_all, c, a, nframes, stream, native);
    default: return launchw_l<10>(mode, in, out, tw_all, c, a, nframes, stream, native);
    }
}

} // namespace intfft
