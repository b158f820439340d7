// intfft_fast1024ux.hip -- UNSCALED (full bit growth) wave kernel, inverse core and FFT->IFFT pair:
// int_ifftNk / int_fft_ifft_pair with FORMAT = 1, DATA_WIDTH = 16, TWDL_WIDTH <= 16, 64 <= N <= 1024, natural
// order in and out (src/vhdl/fft/int_ifftNk.vhd:186-206, src/vhdl/main/int_fft_ifft_pair.vhd:209-280; the
// reference's second testbench, fft_double_test.vhd:83-88, runs this pair with NFFT = 7, FORMAT = 1).
//   inverse alone: int16 in, (16 + L)-bit results in int32 containers (N <= 1024)
//   pair:          int16 in, (16 + 2 L)-bit results; fits int32 containers for N <= 256 (the IFFT DATA_WIDTH is
//                  16 + L: int_fft_ifft_pair.vhd:261)
// The forward half is intfft_u32.hpp's utransform (same code as intfft_fast1024u.hip); the inverse half runs
// the same wave mapping backwards, like intfft_fast1024x.hip:
//   LC  reg = a3..0                  DIT 0, 1 (multiplier-free), 2, 3 (wave-uniform twiddles)
//   LDS transpose (two dword planes) to the "mid" layout; DIT 4; v_permlane16_swap; DIT 5; v_permlane32_swap
//   L1  reg = a9..6, lane = a5..0    DIT 6..9, stored as int2 (512 contiguous bytes per instruction)
// DIT butterfly, unscaled (int_dit2_fly.vhd:142-162, 290-325): the multiplier is fed re/im-swapped,
//   T.im = slice(B.im*wr - B.re*wi),  T.re = slice(B.im*wi + B.re*wr),  X = A + T,  Y = A - T
// at width DTW = W0 + ii (W0 = 16 alone, 16 + L in the pair).  In the pair DTW reaches 29..31 bits, where the
// multiplier is the two-DSP "dbl18" regime (int_cmult_dbl18_dsp48.vhd:163-175): each product is truncated by
// `a` bits before the sum -- evaluated as ((M2 & K) -/+ (M1 & K)) >> (a + b), K = ~(2^a - 1), like
// intfft_wide16.hip; the inverse alone stays in the single-DSP regime (exact sum, chained v_mad_i64_i32).
// Frames that pass the guard-bit vote skip the per-stage width wrap (same bound as intfft_fast1024u.hip: the
// complex magnitude at most doubles per unscaled stage, DIF or DIT).
#include "intfft_u32.hpp"

namespace intfft {

enum { UX_INV = 1, UX_PAIR = 2 };

template <bool WRAP, bool MASKED, bool UNIFORM_W = false>
__device__ __forceinline__ void ufly_dit(int &are, int &aim, int &bre, int &bim, int wr, int wi, const UxStage &s)
{
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi));
    else asm volatile("" : "+v"(wr), "+v"(wi)); // keep the twiddles' sign extension out of loop-invariant hoisting
    unsigned long long xi, xr;
    if (MASKED) {
        const unsigned long long m2i = (unsigned long long)((long long)bim * wr), m1i = (unsigned long long)((long long)bre * wi);
        const unsigned long long m2r = (unsigned long long)((long long)bim * wi), m1r = (unsigned long long)((long long)bre * wr);
        const unsigned long long k = 0xFFFFFFFF00000000ull | s.keep;
        xi = (m2i & k) - (m1i & k); // DO_RE of the swapped feed = T.im
        xr = (m2r & k) + (m1r & k); // DO_IM = T.re
    } else {
        const int nbre = -bre;
        xi = (unsigned long long)((long long)bim * wr + (long long)nbre * wi);
        xr = (unsigned long long)((long long)bim * wi + (long long)bre * wr);
    }
    int tr = (int)__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.sh);
    int ti = (int)__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.sh);
    if (WRAP) {
        tr = __builtin_amdgcn_sbfe(tr, 0, s.w);
        ti = __builtin_amdgcn_sbfe(ti, 0, s.w);
    }
    bre = are - tr;
    bim = aim - ti;
    are += tr;
    aim += ti;
}
// STAGE 0 and even positions of STAGE 1: T = B
__device__ __forceinline__ void ufly_dit_triv(int &are, int &aim, int &bre, int &bim)
{
    const int tr = bre, ti = bim;
    bre = are - tr;
    bim = aim - ti;
    are += tr;
    aim += ti;
}
// odd positions of STAGE 1: T.im = B.re, T.re = B.im >= 0 ? -B.im : ~B.im  (int_dit2_fly.vhd:264-276)
__device__ __forceinline__ void ufly_dit_pj(int &are, int &aim, int &bre, int &bim)
{
    const int tr = (bim >> 31) - bim, ti = bre;
    bre = are - tr;
    bim = aim - ti;
    are += tr;
    aim += ti;
}

// inverse core from LC (reg = a3..0) to L1 (reg = a9..6, lane = a5..0)
template <int L, bool WRAP, bool MASKED, int MAP>
__device__ __forceinline__ void uinverse(int (&re)[16], int (&im)[16], const int (&w9r)[8], const int (&w9i)[8],
                                         const int (&w8r)[4], const int (&w8i)[4], const int (&w7r)[2],
                                         const int (&w7i)[2], int w6r, int w6i, int w5r, int w5i, int w4r, int w4i,
                                         const UConsts &c, const UxArgs &a, u32 *wr_inv, const uint4 *rd_base)
{
    // DIT STAGE 0, 1, 2, 3 on register offsets 1, 2, 4, 8
#pragma unroll
    for (int g = 0; g < 16; g += 2) ufly_dit_triv(re[g], im[g], re[g + 1], im[g + 1]);
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
        ufly_dit_triv(re[g], im[g], re[g + 2], im[g + 2]);
        ufly_dit_pj(re[g + 1], im[g + 1], re[g + 3], im[g + 3]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            ufly_dit<WRAP, MASKED, true>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
#pragma unroll
    for (int r = 0; r < 8; ++r) ufly_dit<WRAP, MASKED, true>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);

    // LC -> mid: element (lane, reg r) -> row = mid lane 32 a9 + 16 a8 + r, column = mid reg (a5 a4 a7 a6)
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        wr_inv[ROWU * r] = (u32)re[r];
        wr_inv[64 * ROWU + ROWU * r] = (u32)im[r];
    }
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 x = rd_base[q], y = rd_base[q + 16 * ROWU];
        re[4 * q + 0] = (int)x.x, re[4 * q + 1] = (int)x.y, re[4 * q + 2] = (int)x.z, re[4 * q + 3] = (int)x.w;
        im[4 * q + 0] = (int)y.x, im[4 * q + 1] = (int)y.y, im[4 * q + 2] = (int)y.z, im[4 * q + 3] = (int)y.w;
    }
    wave_lds_fence();

    // DIT STAGE 4: reg bit 2 = a4; then lane bit 4 <-> reg bit 2
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) ufly_dit<WRAP, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r, w4i, a.st[4]);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uswap16(re[g + j], re[g + j + 4]);
            uswap16(im[g + j], im[g + j + 4]);
        }
    // DIT STAGE 5: reg bit 3 = a5; then lane bit 5 <-> reg bit 3
#pragma unroll
    for (int j = 0; j < 8; ++j) ufly_dit<WRAP, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w5r, w5i, a.st[5]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uswap32(re[j], re[j + 8]);
        uswap32(im[j], im[j + 8]);
    }
    // L1: DIT STAGE 6..9 on register offsets 1, 2, 4, 8 (only the stages of in-frame bits)
    if constexpr (L >= 7) {
#pragma unroll
        for (int g = 0; g < 16; g += 2) ufly_dit<WRAP, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w6r, w6i, a.st[6]);
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int g = 0; g < 16; g += 4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ufly_dit<WRAP, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w7r[j], w7i[j], a.st[7]);
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ufly_dit<WRAP, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w8r[j], w8i[j], a.st[8]);
    }
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ufly_dit<WRAP, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w9r[j], w9i[j], a.st[9]);
    }
}

// four waves per SIMD (the 40 KiB of LDS admit four workgroups): without the hint the N = 256 pair takes 172 VGPRs
// (116 with it, no spills: 289 -> 310 Gsample/s); the other instantiations are unchanged within noise
// NAT (round 4): the inverse core's instantiation for int_ifftNk's own beat orders, selected by `native` (bit 0: HALVES out, bit 1: BITREV in; N >= 128) --
//   BITREV  memory index = core position: an LC lane needs 16 CONSECUTIVE samples (reg = a3..a0); the chunk is loaded in memory order (1 KiB per wave
//           instruction), passed through the wave's idle LDS tile (rows of 16 samples, 20 dwords apart) and read back one row per lane
//   HALVES  a beat (x[i], x[i + N/2]) is the L1 register pair (j0, j0 | 2^(L-7)) of one lane: eight 16-byte stores, 1 KiB per wave instruction
template <int L, int MODE, bool FAST_OK, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FAST_OK ? 4 : 2))) void k_fft1024ux_u32(const u32 *in, int2 *out, const int2 *__restrict__ twt,
                                                       const UConsts c, const UxArgs a, size_t nframes_user, int sh, int native)
{
    static_assert(!NAT || (MODE == UX_INV && L >= 7), "native beat orders: the inverse core alone, N >= 128");
    const bool halves = NAT && (native & 1), bitrev = NAT && (native & 2);
    constexpr int FP = 1 << (10 - L);
    constexpr int MAP = MODE == UX_PAIR ? 1 : 2;
    constexpr bool MASKED = MODE == UX_PAIR;
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
    // forward (mid -> LC) transpose of the pair: natural LC mapping
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    u32 *wr_fwd = lds + ROWU * ((t5 << ulb<L, MAP>(9)) + (t4 << ulb<L, MAP>(8))) + (lane & 15);
    // inverse (LC -> mid): LC lane bit ulb<L, MAP>(k) = a_k
    auto ab = [&](int k) { return (lane >> ulb<L, MAP>(k)) & 1; };
    u32 *wr_inv = lds + ROWU * (32 * ab(9) + 16 * ab(8)) + ((ab(5) << 3) | (ab(4) << 2) | (ab(7) << 1) | ab(6));
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROWU * lane);
    // inverse alone, N < 1024: dwordx4 loads of 4 consecutive X (mirror of intfft_fast1024.hip's short-frame store);
    // while loading (before the swaps) lane bit 5 = a3 and lane bit 4 = a2
    int lane_off = 0, lane_frame = 0;
    if constexpr (L < 10 && MODE == UX_INV) {
        lane_off = ((lane >> 5) & 1) * out_weight<L>(3) + ((lane >> 4) & 1) * out_weight<L>(2);
#pragma unroll
        for (int k = 4; k < 10; ++k) {
            if (k == L - 1 || k == L - 2) continue;
            lane_off += ab(k) * out_weight<L>(k);
            if (k >= L) lane_frame += ab(k) << (k - L);
        }
    }

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const u32 *src = in + f * 1024;
        const bool partial = L < 10 && (f + 1) * FP > nframes_user; // last chunk: absent frames read as 0, not stored
        u32 raw[16];
        if (NAT && bitrev) {
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const v4u *src4 = reinterpret_cast<const v4u *>(src);
            v4u x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { // piece e = 64 i + lane: positions 4 e .. 4 e + 3
                const int e = 64 * i + lane;
                x[i] = v4u{0u, 0u, 0u, 0u};
                if (!partial || f * FP + (size_t)((4 * e) >> L) < nframes_user) x[i] = INTFFT_LD(src4 + e);
            }
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 64 * i + lane;
                *reinterpret_cast<v4u *>(lds + ROWU * (e >> 2) + 4 * (e & 3)) = x[i];
            }
            wave_lds_fence();
            int A = 0; // the index bits a9..a4 this lane carries in LC
#pragma unroll
            for (int k = 4; k < 10; ++k) A |= ab(k) << k;
            const v4u *row = reinterpret_cast<const v4u *>(lds + ROWU * (A >> 4));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4u y = row[q];
                raw[4 * q] = y.x, raw[4 * q + 1] = y.y, raw[4 * q + 2] = y.z, raw[4 * q + 3] = y.w;
            }
            wave_lds_fence();
        } else if (MODE == UX_INV && L < 10) {
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const bool ok = !partial || f * FP + (size_t)lane_frame < nframes_user;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4u x = {0u, 0u, 0u, 0u};
                if (ok)
                    x = INTFFT_LD(
                        reinterpret_cast<const v4u *>(src + lane_off + (q & 1) * out_weight<L>(0) + (q >> 1) * out_weight<L>(1)));
                raw[q] = x.x, raw[q + 8] = x.y, raw[q + 4] = x.z, raw[q + 12] = x.w;
            }
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const auto s = __builtin_amdgcn_permlane16_swap(raw[g + r], raw[g + r + 4], false, false);
                    raw[g + r] = s[0], raw[g + r + 4] = s[1];
                }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const auto s = __builtin_amdgcn_permlane32_swap(raw[r], raw[r + 8], false, false);
                raw[r] = s[0], raw[r + 8] = s[1];
            }
        } else if (MODE == UX_INV) { // LC: raw[r] = X[brev10(n)] = X[64 rev4(r) + lane]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                raw[r] = INTFFT_LD(src + 64 * rr + lane);
            }
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                raw[j] = f * FP + (size_t)((64 * j + lane) >> L) < nframes_user ? src[64 * j + lane] : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) raw[j] = INTFFT_LD(src + 64 * j + lane);
        }
        bool fast = false;
        if (FAST_OK) {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc |= raw[j] + 0x40004000u;
            fast = __builtin_amdgcn_ballot_w64((acc & 0x80008000u) != 0) == 0;
        }
        int re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            re[j] = __builtin_amdgcn_sbfe((int)raw[j], 0, 16);
            im[j] = (int)raw[j] >> 16;
        }
        if (FAST_OK && fast) {
            if (MODE == UX_PAIR)
                utransform<L, false, MAP>(re, im, w9r, w9i, w8r, w8i, w7r, w7i, w6r, w6i, w5r, w5i, w4r, w4i, c, sh, wr_fwd, rd_base);
            uinverse<L, false, MASKED, MAP>(re, im, w9r, w9i, w8r, w8i, w7r, w7i, w6r, w6i, w5r, w5i, w4r, w4i, c, a, wr_inv, rd_base);
        } else {
            if (MODE == UX_PAIR)
                utransform<L, true, MAP>(re, im, w9r, w9i, w8r, w8i, w7r, w7i, w6r, w6i, w5r, w5i, w4r, w4i, c, sh, wr_fwd, rd_base);
            uinverse<L, true, MASKED, MAP>(re, im, w9r, w9i, w8r, w8i, w7r, w7i, w6r, w6i, w5r, w5i, w4r, w4i, c, a, wr_inv, rd_base);
        }
        int2 *dst = out + f * 1024 + lane;
        typedef int v2i __attribute__((ext_vector_type(2)));
        if (NAT && halves) {
            typedef int v4i __attribute__((ext_vector_type(4)));
            v4i *dst4 = reinterpret_cast<v4i *>(out + f * 1024) + lane;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HB = 1 << (L >= 7 ? L - 7 : 0); // register bit that carries a(L-1)
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB);
                const int p0 = 64 * j0;
                const int pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                const v4i y = {re[j0], im[j0], re[j0 | HB], im[j0 | HB]};
                if (!partial || f * FP + (size_t)(p0 >> L) < nframes_user) __builtin_nontemporal_store(y, dst4 + pair);
            }
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) dst[64 * j] = make_int2(re[j], im[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2i y = {re[j], im[j]};
                __builtin_nontemporal_store(y, reinterpret_cast<v2i *>(dst + 64 * j));
            }
        }
    }
}

bool fast1024ux_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly,
                          int in_order, int out_order)
{
    if (!(data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 1 && use_fly == 1)) return false;
    if (direction == 1 && (in_order != 0 || out_order != 0)) // int_ifftNk's own beat orders (BITREV in, HALVES out) and the mixed forms, N >= 128
        return log2n >= 7 && log2n <= 10 && (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2);
    if (in_order != 0 || out_order != 0) return false;
    if (direction == 1) return log2n >= 6 && log2n <= 10;
    if (direction == 2) return log2n >= 6 && log2n <= 8; // 16 + 2 L <= 32 bits
    return false;
}

const char *fast1024ux_kernel_name() { return "k_fft1024ux_u32"; }

template <int L, int MODE, bool FAST_OK, bool NAT = false>
static hipError_t launchux(const u32 *in, int2 *out, const int2 *tw, const UConsts &c, const UxArgs &a, size_t nframes,
                           int sh, hipStream_t stream, int native = 0)
{
    const size_t cap = resident_blocks(kptr(k_fft1024ux_u32<L, MODE, FAST_OK, NAT>), 256, 2);
    const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L);
    const size_t need = (chunks + 3) / 4;
    hipLaunchKernelGGL((k_fft1024ux_u32<L, MODE, FAST_OK, NAT>), dim3((unsigned)(need < cap ? need : cap)), dim3(256), 0, stream,
                       in, out, tw, c, a, nframes, sh, native);
    return hipGetLastError();
}

template <int L>
static hipError_t launchux_l(int direction, bool fast, const u32 *in, int2 *out, const int2 *tw, const UConsts &c,
                             const UxArgs &a, size_t nframes, int sh, hipStream_t stream, int native)
{
    if constexpr (L >= 7) {
        if (direction == 1 && native)
            return fast ? launchux<L, UX_INV, true, true>(in, out, tw, c, a, nframes, sh, stream, native)
                        : launchux<L, UX_INV, false, true>(in, out, tw, c, a, nframes, sh, stream, native);
    }
    if (direction == 1)
        return fast ? launchux<L, UX_INV, true>(in, out, tw, c, a, nframes, sh, stream)
                    : launchux<L, UX_INV, false>(in, out, tw, c, a, nframes, sh, stream);
    if constexpr (L <= 8)
        return fast ? launchux<L, UX_PAIR, true>(in, out, tw, c, a, nframes, sh, stream)
                    : launchux<L, UX_PAIR, false>(in, out, tw, c, a, nframes, sh, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_fast1024ux(int log2n, int direction, int twd, const UxArgs &a, const void *in, void *out,
                             const int2 *tw_all, const int2 *h_tw, size_t nframes, hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const u32 *pin = static_cast<const u32 *>(in);
    int2 *pout = static_cast<int2 *>(out);
    switch (log2n) {
    case 6: return launchux_l<6>(direction, allow_fast, pin, pout, tw_all, c, a, nframes, twd - 1, stream, native);
    case 7: return launchux_l<7>(direction, allow_fast, pin, pout, tw_all, c, a, nframes, twd - 1, stream, native);
    case 8: return launchux_l<8>(direction, allow_fast, pin, pout, tw_all, c, a, nframes, twd - 1, stream, native);
    case 9: return launchux_l<9>(direction, allow_fast, pin, pout, tw_all, c, a, nframes, twd - 1, stream, native);
    default: return launchux_l<10>(direction, allow_fast, pin, pout, tw_all, c, a, nframes, twd - 1, stream, native);
    }
}

} // namespace intfft
