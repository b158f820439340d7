// intfft_w32inv.hip -- general-width inverse kernels: int_ifftNk for 64 <= N <= 4096 with any DATA_WIDTH / TWDL_WIDTH /
// FORMAT / RNDMODE whose widths stay within 32 bits, natural order in and out (e.g. the unscaled 16-bit inverse at
// N = 2048 / 4096, 12-bit or 24-bit scaled inverses).  Mirrors of intfft_fastw32.hip (wave kernel, N <= 1024) and
// intfft_fast4096w.hip (block kernel, N = 2048 / 4096) with the parameterised DIT butterflies of intfft_u32.hpp:
//   wave   LC (reg = a3..0, X[brev_L(n)] loaded with the bit reversal in the addressing): DIT 0..3; LDS transpose;
//          DIT 4; v_permlane16_swap; DIT 5; v_permlane32_swap; L1 (reg = a9..6): DIT 6..L-1; coalesced store
//   block  LC: DIT 0..3; transpose; LB (reg = n7..4): DIT 4..7; transpose; LA (reg = n11..8): DIT 8..L-1; coalesced store
// (DATA_WIDTH = 16 with TWDL_WIDTH <= 16 has the tuned kernels intfft_fast1024x.hip, intfft_fast4096.hip and
// intfft_fast1024ux.hip.)
#include "intfft_u32.hpp"

namespace intfft {

constexpr int ROW4X = 20;
constexpr int PLANE4X = 256 * ROW4X;
template <int L> __host__ __device__ constexpr int lcx_bit(int k) { return k < L ? (L - 1) - k : (L - 4) + (k - L); }
__device__ __forceinline__ constexpr int rev4x(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// DIT 0..3 on register offsets 1, 2, 4, 8 with the wave-uniform STAGE 2 / 3 twiddles
template <int MODE, bool MASKED>
__device__ __forceinline__ void ground_c_dit(int (&re)[16], int (&im)[16], const UConsts &c, const W32Args &a)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly_dit_triv<MODE, false>(re[g], im[g], re[g + 1], im[g + 1], a.st[0]);
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
        gfly_dit_triv<MODE, false>(re[g], im[g], re[g + 2], im[g + 2], a.st[1]);
        gfly_dit_triv<MODE, true>(re[g + 1], im[g + 1], re[g + 3], im[g + 3], a.st[1]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            gfly_dit<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
#pragma unroll
    for (int r = 0; r < 8; ++r) gfly_dit<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
}

// ---- wave kernel, 64 <= N <= 1024 ---------------------------------------------------------------------------------------
// NAT (round 4): the instantiation for int_ifftNk's own beat orders (`native` bit 0: HALVES out, bit 1: BITREV in; N >= 128), as in
// intfft_fast1024ux.hip: BITREV order = the core position, 16 consecutive ones per LC lane -- the chunk is loaded in memory order (1 KiB per wave
// instruction), passed through the wave's idle LDS tile (padded rows of 16 samples) and read back one row per lane; HALVES beats = one 8- / 16-byte
// store of the L1 register pair (j0, j0 | 2^(L-7))
template <int L, int MODE, bool MASKED, bool NAT = false>
__global__ __launch_bounds__(256) void k_ifft1024_w32(const void *in, void *out, const int2 *__restrict__ twt, const UConsts c,
                                                      const W32Args a, size_t nframes_user, int native)
{
    static_assert(!NAT || L >= 7, "native beat orders: N >= 128");
    const bool halves = NAT && (native & 1), bitrev = NAT && (native & 2);
    constexpr int FP = 1 << (10 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP;
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
    // LC lane mapping lane_bit_u<L> (intfft_fast1024u.hip); LC -> mid transpose: row = mid lane 32 a9 + 16 a8 + r,
    // column = mid register (a5 a4 a7 a6)
    auto ab = [&](int k) { return (lane >> lane_bit_u<L>(k)) & 1; };
    u32 *wr_inv = lds + ROWU * (32 * ab(9) + 16 * ab(8)) + ((ab(5) << 3) | (ab(4) << 2) | (ab(7) << 1) | ab(6));
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROWU * lane);
    // N < 1024: every lane loads pairs of consecutive X (mirror of the forward kernel's one-swap store);
    // while loading (before the swap) lane bit 5 = a3
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

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user;
        int re[16], im[16];
        // ---- load X[brev_L(n)] into LC ----
        if (NAT && bitrev) {
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            int A = 0; // the index bits a9..a4 this lane carries in LC
#pragma unroll
            for (int k = 4; k < 10; ++k) A |= ab(k) << k;
            if (a.in16) { // 4 B per sample: pieces of 4 samples, rows of 16 dwords 20 apart
                const v4u *src4 = reinterpret_cast<const v4u *>(static_cast<const u32 *>(in) + f * 1024);
                v4u x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 64 * i + lane;
                    x[i] = v4u{0u, 0u, 0u, 0u};
                    if (!partial || f * FP + (size_t)((4 * e) >> L) < nframes_user) x[i] = INTFFT_LD(src4 + e);
                }
                wave_lds_fence();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 64 * i + lane;
                    *reinterpret_cast<v4u *>(lds + 20 * (e >> 2) + 4 * (e & 3)) = x[i];
                }
                wave_lds_fence();
                const v4u *row = reinterpret_cast<const v4u *>(lds + 20 * (A >> 4));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4u y = row[q];
                    const u32 raw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        re[4 * q + t] = (int)(raw[t] << a.in_sh) >> a.in_sh, im[4 * q + t] = (int)(raw[t] << (a.in_sh - 16)) >> a.in_sh;
                }
            } else { // 8 B per sample: pieces of 2 samples, rows of 32 dwords 36 apart
                const v4u *src4 = reinterpret_cast<const v4u *>(static_cast<const int2 *>(in) + f * 1024);
                v4u x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 64 * i + lane;
                    x[i] = v4u{0u, 0u, 0u, 0u};
                    if (!partial || f * FP + (size_t)((2 * e) >> L) < nframes_user) x[i] = INTFFT_LD(src4 + e);
                }
                wave_lds_fence();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 64 * i + lane;
                    *reinterpret_cast<v4u *>(lds + 36 * (e >> 3) + 4 * (e & 7)) = x[i];
                }
                wave_lds_fence();
                const v4u *row = reinterpret_cast<const v4u *>(lds + 36 * (A >> 4));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4u y = row[q];
                    re[2 * q] = (int)(y.x << a.in_sh) >> a.in_sh, im[2 * q] = (int)(y.y << a.in_sh) >> a.in_sh;
                    re[2 * q + 1] = (int)(y.z << a.in_sh) >> a.in_sh, im[2 * q + 1] = (int)(y.w << a.in_sh) >> a.in_sh;
                }
            }
            wave_lds_fence();
        } else if constexpr (L < 10) {
            const bool ok = !partial || f * FP + (size_t)lane_frame < nframes_user;
            // (the container test stays outside the unrolled loops: one batch of loads in flight, not 16 round trips)
            if (a.in16) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                v2u x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const size_t off = f * 1024 + lane_off + (q & 1) * out_weight<L>(0) + ((q >> 1) & 1) * out_weight<L>(1) +
                                       (q >> 2) * out_weight<L>(2);
                    x[q] = v2u{0u, 0u};
                    if (ok) x[q] = INTFFT_LD(reinterpret_cast<const v2u *>(static_cast<const u32 *>(in) + off));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    re[q] = (int)(x[q].x << a.in_sh) >> a.in_sh, im[q] = (int)(x[q].x << (a.in_sh - 16)) >> a.in_sh;
                    re[q + 8] = (int)(x[q].y << a.in_sh) >> a.in_sh, im[q + 8] = (int)(x[q].y << (a.in_sh - 16)) >> a.in_sh;
                }
            } else {
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const size_t off = f * 1024 + lane_off + (q & 1) * out_weight<L>(0) + ((q >> 1) & 1) * out_weight<L>(1) +
                                       (q >> 2) * out_weight<L>(2);
                    x[q] = v4i{0, 0, 0, 0};
                    if (ok) x[q] = INTFFT_LD(reinterpret_cast<const v4i *>(static_cast<const int2 *>(in) + off));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    re[q] = (int)((u32)x[q].x << a.in_sh) >> a.in_sh, im[q] = (int)((u32)x[q].y << a.in_sh) >> a.in_sh;
                    re[q + 8] = (int)((u32)x[q].z << a.in_sh) >> a.in_sh, im[q + 8] = (int)((u32)x[q].w << a.in_sh) >> a.in_sh;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uswap32(re[r], re[r + 8]);
                uswap32(im[r], im[r + 8]);
            }
        } else {
            if (a.in16) {
                const u32 *src = static_cast<const u32 *>(in) + f * 1024 + lane;
                u32 raw[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) raw[r] = INTFFT_LD(src + 64 * rev4x(r));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    re[r] = (int)(raw[r] << a.in_sh) >> a.in_sh, im[r] = (int)(raw[r] << (a.in_sh - 16)) >> a.in_sh;
            } else {
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + f * 1024 + lane);
                v2i x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = INTFFT_LD(src + 64 * rev4x(r));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    re[r] = (int)((u32)x[r].x << a.in_sh) >> a.in_sh, im[r] = (int)((u32)x[r].y << a.in_sh) >> a.in_sh;
            }
        }
        ground_c_dit<MODE, MASKED>(re, im, c, a);
        // ---- LC -> mid ----
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
        // DIT 4 (reg bit 2 = a4), lane bit 4 <-> reg bit 2, DIT 5 (reg bit 3 = a5), lane bit 5 <-> reg bit 3
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r, w4i, a.st[4]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uswap16(re[g + j], re[g + j + 4]);
                uswap16(im[g + j], im[g + j + 4]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) gfly_dit<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w5r, w5i, a.st[5]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uswap32(re[j], re[j + 8]);
            uswap32(im[j], im[j + 8]);
        }
        // L1: DIT 6..L-1
        if constexpr (L >= 7) {
#pragma unroll
            for (int g = 0; g < 16; g += 2) gfly_dit<MODE, false, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w6r, w6i, a.st[6]);
        }
        if constexpr (L >= 8) {
#pragma unroll
            for (int g = 0; g < 16; g += 4)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w7r[j], w7i[j], a.st[7]);
        }
        if constexpr (L >= 9) {
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w8r[j], w8i[j], a.st[8]);
        }
        if constexpr (L >= 10) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gfly_dit<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w9r[j], w9i[j], a.st[9]);
        }
        // ---- store (L1 layout: natural order, coalesced) ----
        if (NAT && halves) {
            constexpr int HB = 1 << (L >= 7 ? L - 7 : 0); // register bit that carries a(L-1)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB);
                const int p0 = 64 * j0;
                const int pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                if (partial && !(f * FP + (size_t)(p0 >> L) < nframes_user)) continue;
                if (a.out16) {
                    typedef u32 v2u __attribute__((ext_vector_type(2)));
                    const v2u y = {((u32)re[j0] & 0xFFFFu) | ((u32)im[j0] << 16), ((u32)re[j0 | HB] & 0xFFFFu) | ((u32)im[j0 | HB] << 16)};
                    __builtin_nontemporal_store(y, reinterpret_cast<v2u *>(static_cast<u32 *>(out) + f * 1024) + lane + pair);
                } else {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    const v4i y = {re[j0], im[j0], re[j0 | HB], im[j0 | HB]};
                    __builtin_nontemporal_store(y, reinterpret_cast<v4i *>(static_cast<int2 *>(out) + f * 1024) + lane + pair);
                }
            }
        } else if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + f * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user)
                    __builtin_nontemporal_store(((u32)re[j] & 0xFFFFu) | ((u32)im[j] << 16), dst + 64 * j);
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + f * 1024 + lane);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) {
                    const v2i y = {re[j], im[j]};
                    __builtin_nontemporal_store(y, dst + 64 * j);
                }
        }
    }
}

// ---- block kernel, N = 2048 / 4096 ---------------------------------------------------------------------------------------
template <int MODE, bool MASKED, int L, int S0>
__device__ __forceinline__ void ground_dit(int (&re)[16], int (&im)[16], const int (&w8r)[8], const int (&w8i)[8],
                                           const int (&w4r)[4], const int (&w4i)[4], const int (&w2r)[2], const int (&w2i)[2],
                                           int w1r, int w1i, const W32Args &a)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly_dit<MODE, false, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w1r, w1i, a.st[S0]);
#pragma unroll
    for (int g = 0; g < 16; g += 4)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w2r[j], w2i[j], a.st[S0 + 1]);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r[j], w4i[j], a.st[S0 + 2]);
    if constexpr (S0 + 3 < L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gfly_dit<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w8r[j], w8i[j], a.st[S0 + 3]);
    }
}

// NAT (round 4): int_ifftNk's own beat orders (`native` bit 0: HALVES out, bit 1: BITREV in): the chunk loaded in memory order and handed to the LC
// threads (16 consecutive core positions each) through padded rows of the transpose region; HALVES beats as 8- / 16-byte stores of LA register pairs
template <int L, int MODE, bool MASKED, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_ifft4096_w32(const void *in, void *out, const int2 *__restrict__ twt, const UConsts c, const W32Args a,
                    size_t nframes_user, int native)
{
    const bool halves = NAT && (native & 1), bitrev = NAT && (native & 2);
    static_assert(L == 11 || L == 12, "block kernel: N = 2048 or 4096");
    constexpr int FP = 1 << (12 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP;
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANE4X];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;

    int a8r[8] = {}, a8i[8] = {}, a4r[4], a4i[4], a2r[2], a2i[2], a1r, a1i;
    int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
    {
        int2 w;
        if constexpr (L >= 12) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w = twt[2047 + 256 * j + tid], a8r[j] = w.x, a8i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) w = twt[1023 + 256 * j + tid], a4r[j] = w.x, a4i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 2; ++j) w = twt[511 + 256 * j + tid], a2r[j] = w.x, a2i[j] = w.y;
        w = twt[255 + tid], a1r = w.x, a1i = w.y;
#pragma unroll
        for (int j = 0; j < 8; ++j) w = twt[127 + 16 * j + lo4], b8r[j] = w.x, b8i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) w = twt[63 + 16 * j + lo4], b4r[j] = w.x, b4i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 2; ++j) w = twt[31 + 16 * j + lo4], b2r[j] = w.x, b2i[j] = w.y;
        w = twt[15 + lo4], b1r = w.x, b1i = w.y;
    }
    // LC thread t'' carries n_k on bit lcx_bit<L>(k); LC -> LB: row = LB thread 16 (n11..8) + r, column = n7..4
    auto nb = [&](int k) { return (tid >> lcx_bit<L>(k)) & 1; };
    const int lb_hi = nb(8) | (nb(9) << 1) | (nb(10) << 2) | (nb(11) << 3), lb_reg = nb(4) | (nb(5) << 1) | (nb(6) << 2) | (nb(7) << 3);
    // blocks of 16 rows 316 dwords apart and columns in the order n6 + 2 n7 + 4 n4 + 8 n5: the wave's 64 writes (n11..8, n7..6
    // in its lane bits) go to 64 different banks -- with blocks at 16 * ROW4X = 0 mod 64 they shared four (intfft_fast4096.hip)
    constexpr int BLK_CB = 316;
    u32 *const w_cb = lds + BLK_CB * lb_hi + (lb_reg >> 2) + 4 * (lb_reg & 3);
    // LB -> LA: element (thread (n11..8 = hi4, n3..0 = lo4), reg n7..4) -> row n7..0 = 16 j' + lo4, column n11..8 = hi4
    u32 *const w_ba = lds + ROW4X * lo4 + hi4;
    const uint4 *const rd0 = reinterpret_cast<const uint4 *>(lds + ROW4X * tid);
    const uint4 *const rd1 = reinterpret_cast<const uint4 *>(lds + PLANE4X + ROW4X * tid);
    const uint4 *const rc0 = reinterpret_cast<const uint4 *>(lds + BLK_CB * hi4 + ROW4X * lo4);
    const uint4 *const rc1 = reinterpret_cast<const uint4 *>(lds + PLANE4X + BLK_CB * hi4 + ROW4X * lo4);
    int lc_off = 0, lc_frame = 0;
#pragma unroll
    for (int k = 4; k < 12; ++k) {
        lc_off += nb(k) * (k >= L ? (1 << k) : (1 << (L - 1 - k)));
        if (k >= L) lc_frame += nb(k) << (k - L);
    }
    auto transpose_read_cb = [&](int (&re)[16], int (&im)[16]) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rc0[q], y = rc1[q];
            re[q] = (int)x.x, re[q + 4] = (int)x.y, re[q + 8] = (int)x.z, re[q + 12] = (int)x.w;
            im[q] = (int)y.x, im[q + 4] = (int)y.y, im[q + 8] = (int)y.z, im[q + 12] = (int)y.w;
        }
        __syncthreads();
    };
    auto transpose_read = [&](int (&re)[16], int (&im)[16]) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd0[q], y = rd1[q];
            re[4 * q + 0] = (int)x.x, re[4 * q + 1] = (int)x.y, re[4 * q + 2] = (int)x.z, re[4 * q + 3] = (int)x.w;
            im[4 * q + 0] = (int)y.x, im[4 * q + 1] = (int)y.y, im[4 * q + 2] = (int)y.z, im[4 * q + 3] = (int)y.w;
        }
        __syncthreads();
    };

    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const bool partial = L < 12 && (f + 1) * FP > nframes_user;
        const bool lc_ok = !partial || f * FP + (size_t)lc_frame < nframes_user;
        int re[16], im[16];
        // LC: X[brev_L(n)] of the thread's frame (container test outside the unrolled loops)
        if (NAT && bitrev) {
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            int A = 0; // the index bits n11..n4 this thread carries in LC
#pragma unroll
            for (int k = 4; k < 12; ++k) A |= nb(k) << k;
            if (a.in16) {
                const v4u *src4 = reinterpret_cast<const v4u *>(static_cast<const u32 *>(in) + f * 4096);
                v4u x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 256 * i + tid; // samples 4 e .. 4 e + 3
                    x[i] = v4u{0u, 0u, 0u, 0u};
                    if (!partial || f * FP + (size_t)((4 * e) >> L) < nframes_user) x[i] = INTFFT_LD(src4 + e);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 256 * i + tid;
                    *reinterpret_cast<v4u *>(lds + 20 * (e >> 2) + 4 * (e & 3)) = x[i];
                }
                __syncthreads();
                const v4u *row = reinterpret_cast<const v4u *>(lds + 20 * (A >> 4));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4u y = row[q];
                    const u32 raw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        re[4 * q + t] = (int)(raw[t] << a.in_sh) >> a.in_sh, im[4 * q + t] = (int)(raw[t] << (a.in_sh - 16)) >> a.in_sh;
                }
            } else {
                const v4u *src4 = reinterpret_cast<const v4u *>(static_cast<const int2 *>(in) + f * 4096);
                v4u x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 256 * i + tid; // samples 2 e, 2 e + 1
                    x[i] = v4u{0u, 0u, 0u, 0u};
                    if (!partial || f * FP + (size_t)((2 * e) >> L) < nframes_user) x[i] = INTFFT_LD(src4 + e);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 256 * i + tid;
                    *reinterpret_cast<v4u *>(lds + 36 * (e >> 3) + 4 * (e & 7)) = x[i];
                }
                __syncthreads();
                const v4u *row = reinterpret_cast<const v4u *>(lds + 36 * (A >> 4));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4u y = row[q];
                    re[2 * q] = (int)(y.x << a.in_sh) >> a.in_sh, im[2 * q] = (int)(y.y << a.in_sh) >> a.in_sh;
                    re[2 * q + 1] = (int)(y.z << a.in_sh) >> a.in_sh, im[2 * q + 1] = (int)(y.w << a.in_sh) >> a.in_sh;
                }
            }
            __syncthreads(); // the region takes the LC -> LB rows next
        } else if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + f * 4096 + lc_off;
            u32 raw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) raw[r] = lc_ok ? INTFFT_LD(src + (rev4x(r) << (L - 4))) : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                re[r] = (int)(raw[r] << a.in_sh) >> a.in_sh, im[r] = (int)(raw[r] << (a.in_sh - 16)) >> a.in_sh;
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + f * 4096 + lc_off);
            v2i x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = v2i{0, 0};
                if (lc_ok) x[r] = INTFFT_LD(src + (rev4x(r) << (L - 4)));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                re[r] = (int)((u32)x[r].x << a.in_sh) >> a.in_sh, im[r] = (int)((u32)x[r].y << a.in_sh) >> a.in_sh;
        }
        ground_c_dit<MODE, MASKED>(re, im, c, a);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            w_cb[ROW4X * r] = (u32)re[r];
            w_cb[PLANE4X + ROW4X * r] = (u32)im[r];
        }
        transpose_read_cb(re, im); // LB: regs = n7..4
        ground_dit<MODE, MASKED, 12, 4>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            w_ba[ROW4X * 16 * j] = (u32)re[j];
            w_ba[PLANE4X + ROW4X * 16 * j] = (u32)im[j];
        }
        transpose_read(re, im); // LA: regs = n11..8, thread = n7..0
        ground_dit<MODE, MASKED, L, 8>(re, im, a8r, a8i, a4r, a4i, a2r, a2i, a1r, a1i, a);
        if (NAT && halves) {
            constexpr int HB = 1 << (L - 9); // LA register bit that carries n(L-1)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB);
                const int p0 = 256 * j0;
                const int pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                if (partial && !(f * FP + (size_t)(p0 >> L) < nframes_user)) continue;
                if (a.out16) {
                    typedef u32 v2u __attribute__((ext_vector_type(2)));
                    const v2u y = {((u32)re[j0] & 0xFFFFu) | ((u32)im[j0] << 16), ((u32)re[j0 | HB] & 0xFFFFu) | ((u32)im[j0 | HB] << 16)};
                    __builtin_nontemporal_store(y, reinterpret_cast<v2u *>(static_cast<u32 *>(out) + f * 4096) + tid + pair);
                } else {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    const v4i y = {re[j0], im[j0], re[j0 | HB], im[j0 | HB]};
                    __builtin_nontemporal_store(y, reinterpret_cast<v4i *>(static_cast<int2 *>(out) + f * 4096) + tid + pair);
                }
            }
        } else if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + f * 4096 + tid;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || f * FP + (size_t)((256 * j + tid) >> L) < nframes_user)
                    __builtin_nontemporal_store(((u32)re[j] & 0xFFFFu) | ((u32)im[j] << 16), dst + 256 * j);
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + f * 4096 + tid);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || f * FP + (size_t)((256 * j + tid) >> L) < nframes_user) {
                    const v2i y = {re[j], im[j]};
                    __builtin_nontemporal_store(y, dst + 256 * j);
                }
        }
    }
}

bool w32inv_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                      int out_order)
{
    if (!(log2n >= 6 && log2n <= 12 && data_width >= 2 && data_width + format * log2n <= 32 && twdl_width >= 4 && twdl_width <= 26 && direction == 1 &&
          use_fly == 1))
        return false;
    if (in_order == 0 && out_order == 0) return true;
    // int_ifftNk's own beat orders (BITREV in, HALVES out) and the mixed forms, N = 128 .. 4096
    return log2n >= 7 && (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2);
}

const char *w32inv_kernel_name(int log2n) { return log2n > 10 ? "k_ifft4096_w32" : "k_ifft1024_w32"; }

template <int L, int MODE, bool MASKED>
static hipError_t launchxi(const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a, size_t nframes,
                           hipStream_t stream, int native)
{
    if constexpr (L > 10) {
        if (native) {
            const size_t capn = resident_blocks(kptr(k_ifft4096_w32<L, MODE, MASKED, true>), 256, 2);
            const size_t chunks = (nframes + ((size_t)1 << (12 - L)) - 1) >> (12 - L);
            hipLaunchKernelGGL((k_ifft4096_w32<L, MODE, MASKED, true>), dim3((unsigned)(chunks < capn ? chunks : capn)), dim3(256), 0, stream, in, out, tw, c,
                               a, nframes, native);
            return hipGetLastError();
        }
    }
    if constexpr (L >= 7 && L <= 10) {
        if (native) {
            const size_t capn = resident_blocks(kptr(k_ifft1024_w32<L, MODE, MASKED, true>), 256, 2);
            const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L);
            const size_t need = (chunks + 3) / 4;
            hipLaunchKernelGGL((k_ifft1024_w32<L, MODE, MASKED, true>), dim3((unsigned)(need < capn ? need : capn)), dim3(256), 0, stream, in, out, tw, c, a,
                               nframes, native);
            return hipGetLastError();
        }
    }
    constexpr bool BLOCK = L > 10;
    const void *kernel;
    if constexpr (BLOCK) kernel = kptr(k_ifft4096_w32<L, MODE, MASKED>);
    else kernel = kptr(k_ifft1024_w32<L, MODE, MASKED>);
    const size_t cap = resident_blocks(kernel, 256, 2);
    if constexpr (BLOCK) {
        const size_t chunks = (nframes + ((size_t)1 << (12 - L)) - 1) >> (12 - L);
        hipLaunchKernelGGL((k_ifft4096_w32<L, MODE, MASKED>), dim3((unsigned)(chunks < cap ? chunks : cap)), dim3(256), 0, stream,
                           in, out, tw, c, a, nframes, 0);
    } else {
        const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L);
        const size_t need = (chunks + 3) / 4;
        hipLaunchKernelGGL((k_ifft1024_w32<L, MODE, MASKED>), dim3((unsigned)(need < cap ? need : cap)), dim3(256), 0, stream, in,
                           out, tw, c, a, nframes, 0);
    }
    return hipGetLastError();
}

template <int L>
static hipError_t launchxi_l(int mode, const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a,
                             size_t nframes, hipStream_t stream, int native)
{
    if (a.masked) {
        switch (mode) {
        case W_TRUNC: return launchxi<L, W_TRUNC, true>(in, out, tw, c, a, nframes, stream, native);
        case W_ROUND: return launchxi<L, W_ROUND, true>(in, out, tw, c, a, nframes, stream, native);
        default: return launchxi<L, W_UNSCALED, true>(in, out, tw, c, a, nframes, stream, native);
        }
    }
    switch (mode) {
    case W_TRUNC: return launchxi<L, W_TRUNC, false>(in, out, tw, c, a, nframes, stream, native);
    case W_ROUND: return launchxi<L, W_ROUND, false>(in, out, tw, c, a, nframes, stream, native);
    default: return launchxi<L, W_UNSCALED, false>(in, out, tw, c, a, nframes, stream, native);
    }
}

hipError_t launch_w32inv(int log2n, int mode, const W32Args &a, const void *in, void *out, const int2 *tw_all,
                         const int2 *h_tw, size_t nframes, hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    switch (log2n) {
    case 6: return launchxi_l<6>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 7: return launchxi_l<7>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 8: return launchxi_l<8>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 9: return launchxi_l<9>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 10: return launchxi_l<10>(mode, in, out, tw_all, c, a, nframes, stream, native);
    case 11: return launchxi_l<11>(mode, in, out, tw_all, c, a, nframes, stream, native);
    default: return launchxi_l<12>(mode, in, out, tw_all, c, a, nframes, stream, native);
    }
}

} // namespace intfft
