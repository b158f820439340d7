// intfft_bigw.hip -- general-width three-pass kernels for N = 2^13 .. 2^16: int_fftNk with any DATA_WIDTH / TWDL_WIDTH /
// FORMAT / RNDMODE whose widths stay within 32 bits (e.g. the unscaled 16-bit transform up to N = 65536: 32-bit
// results), natural order in and out.  The pass structure of intfft_big20.hip on unpacked int32 registers with the
// parameterised butterflies of intfft_u32.hpp; the plan scratch holds int32 (re, im) pairs at the core index:
//   pass 1  STAGE L-1..12  groups of 2^(16-L) frames form a virtual 2^16-point frame (frame-number stages skipped);
//                          regs = n15..12 at stride 4096, thread = 512 consecutive n (4 KiB runs), no LDS
//   pass 2  STAGE 11..4    every 4096-point block in place: LA (regs n11..8) -> LDS -> LB (regs n7..4) -> LDS -> LA
//   pass 3  STAGE 3..0     tile = 256 values of n(L-1)..n(L-8) x 32 consecutive n, LDS transpose to regs n3..0 with
//                          thread = (n4, rev8(top 8 bits)): natural-order stores in 1-2 KiB runs
// (DATA_WIDTH = 16 scaled-truncate with TWDL_WIDTH <= 16 has the packed kernels of intfft_big20.hip.)
// multi-pass kernels: non-temporal loads measure 4-14 % faster here (the single-pass kernels gain 4-30 % from PLAIN loads): intfft_device.hpp
#define INTFFT_NT_LOADS 1
#include "intfft_u32.hpp"

namespace intfft {

constexpr int ROWG = 17;              // LDS row stride in dwords (odd: conflict-free rows and columns)
constexpr int PLANEG2 = 256 * ROWG;   // pass 2: 256 threads
constexpr int PLANEG3 = 512 * ROWG;   // pass 3: 512 threads
__device__ __forceinline__ constexpr int rev4g(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// four DIF stages s0+3 .. s0 on register offsets 8, 4, 2, 1; only the last NS of them
template <int MODE, bool MASKED, int NS, int S0>
__device__ __forceinline__ void gstages(int (&re)[16], int (&im)[16], const int (&w8r)[8], const int (&w8i)[8],
                                        const int (&w4r)[4], const int (&w4i)[4], const int (&w2r)[2], const int (&w2i)[2],
                                        int w1r, int w1i, const W32Args &a)
{
    if constexpr (NS >= 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gfly<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w8r[j], w8i[j], a.st[S0 + 3]);
    }
    if constexpr (NS >= 3) {
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r[j], w4i[j], a.st[S0 + 2]);
    }
    if constexpr (NS >= 2) {
#pragma unroll
        for (int g = 0; g < 16; g += 4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w2r[j], w2i[j], a.st[S0 + 1]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly<MODE, false, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w1r, w1i, a.st[S0]);
}

// ---- pass 1 ----------------------------------------------------------------------------------------------------
template <int L, int MODE, bool MASKED>
__global__ __launch_bounds__(512) void k_bigw_p1(const void *in, int2 *scr, const int2 *__restrict__ twt, const W32Args a,
                                                 size_t nframes_user, unsigned groups)
{
    static_assert(L >= 13 && L <= 16, "one-round pass 1");
    constexpr int NS = L - 12, G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G; // virtual 2^16-point frames
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups; // chunk 0..7
    const unsigned lfull = chunk * 512 + threadIdx.x;                        // n11..0
    int w8r[8] = {}, w8i[8] = {}, w4r[4] = {}, w4i[4] = {}, w2r[2] = {}, w2i[2] = {}, w1r, w1i;
    {
        int2 w;
        if constexpr (NS >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w = twt[(1u << 15) - 1u + lfull + (unsigned)j * 4096u], w8r[j] = w.x, w8i[j] = w.y;
        }
        if constexpr (NS >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w = twt[(1u << 14) - 1u + lfull + (unsigned)j * 4096u], w4r[j] = w.x, w4i[j] = w.y;
        }
        if constexpr (NS >= 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) w = twt[(1u << 13) - 1u + lfull + (unsigned)j * 4096u], w2r[j] = w.x, w2i[j] = w.y;
        }
        w = twt[(1u << 12) - 1u + lfull], w1r = w.x, w1i = w.y;
    }
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const bool partial = L < 16 && (frame + 1) * G > nframes_user; // last group: absent frames read as 0, not stored
        int re[16], im[16];
        if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + frame * 65536 + lfull;
            u32 raw[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                raw[j] = (!partial || frame * G + (size_t)(j >> (L - 12)) < nframes_user) ? INTFFT_LD(src + ((size_t)j << 12)) : 0u;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                re[j] = (int)(raw[j] << a.in_sh) >> a.in_sh, im[j] = (int)(raw[j] << (a.in_sh - 16)) >> a.in_sh;
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + frame * 65536 + lfull);
            v2i x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                x[j] = v2i{0, 0};
                if (!partial || frame * G + (size_t)(j >> (L - 12)) < nframes_user) x[j] = INTFFT_LD(src + ((size_t)j << 12));
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                re[j] = (int)((u32)x[j].x << a.in_sh) >> a.in_sh, im[j] = (int)((u32)x[j].y << a.in_sh) >> a.in_sh;
        }
        gstages<MODE, MASKED, NS, 12>(re, im, w8r, w8i, w4r, w4i, w2r, w2i, w1r, w1i, a);
        int2 *dst = scr + frame * 65536 + lfull;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (!partial || frame * G + (size_t)(j >> (L - 12)) < nframes_user) dst[(size_t)j << 12] = make_int2(re[j], im[j]);
    }
}

// ---- pass 2 ----------------------------------------------------------------------------------------------------
template <int MODE, bool MASKED>
__global__ __launch_bounds__(256) void k_bigw_p2(int2 *scr, const int2 *__restrict__ twt, const W32Args a, size_t nblocks4k)
{
    __shared__ u32 lds[2 * PLANEG2];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    int a8r[8], a8i[8], a4r[4], a4i[4], a2r[2], a2i[2], a1r, a1i;
    int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
    {
        int2 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w = twt[2047 + 256 * j + tid], a8r[j] = w.x, a8i[j] = w.y;
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
    // LA <-> LB (both directions): element (thread x, reg y) -> row 16 y + x3..0, column x7..4; thread t reads row t
    u32 *const wr = lds + ROWG * lo4 + hi4;
    const u32 *const rd = lds + ROWG * tid;
    auto transpose = [&](int (&re)[16], int (&im)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            wr[ROWG * 16 * j] = (u32)re[j];
            wr[PLANEG2 + ROWG * 16 * j] = (u32)im[j];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)rd[r], im[r] = (int)rd[PLANEG2 + r];
        __syncthreads();
    };
    for (size_t b = blockIdx.x; b < nblocks4k; b += gridDim.x) {
        int2 *p = scr + b * 4096;
        int re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int2 x = p[256 * j + tid]; // LA: regs = n11..8, thread = n7..0
            re[j] = x.x, im[j] = x.y;
        }
        gstages<MODE, MASKED, 4, 8>(re, im, a8r, a8i, a4r, a4i, a2r, a2i, a1r, a1i, a);
        transpose(re, im); // LB: regs = n7..4, thread = (n11..8, n3..0)
        gstages<MODE, MASKED, 4, 4>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
        transpose(re, im); // LA again: coalesced store
#pragma unroll
        for (int r = 0; r < 16; ++r) p[256 * r + tid] = make_int2(re[r], im[r]);
    }
}

// ---- pass 3 ----------------------------------------------------------------------------------------------------
template <int MODE, bool MASKED>
__global__ __launch_bounds__(512) void k_bigw_p3(const int2 *scr, void *out, const UConsts c, const W32Args a, int L)
{
    __shared__ u32 lds[PLANEG3]; // one 34 KiB plane, used for re then im (two planes would exceed 64 KiB)
    const int tid = threadIdx.x, e = tid & 31, px = tid >> 5; // e = n4..0, px = n(L-5)..n(L-8)
    const size_t frame = blockIdx.x >> (L - 13); // frame-major: the tiles of one frame are neighbouring blocks
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u)); // n(L-9)..5
    const int2 *src = scr + (frame << L) + mid * 32 + e;
    int re[16], im[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int2 x = src[(size_t)(16 * j + px) << (L - 8)]; // reg j = n(L-1)..n(L-4)
        re[j] = x.x, im[j] = x.y;
    }
    // transpose -> regs = n3..0, thread = (n4, rev8(top 8 bits)); rev8(16 j + px) = 16 rev4(px) + rev4(j)
    {
        u32 *w = lds + ROWG * ((e >> 4) * 256 + 16 * rev4g(px)) + (e & 15);
#pragma unroll
        for (int j = 0; j < 16; ++j) w[ROWG * rev4g(j)] = (u32)re[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)lds[ROWG * tid + r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) w[ROWG * rev4g(j)] = (u32)im[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) im[r] = (int)lds[ROWG * tid + r];
    }
    // stages 3, 2 (uniform twiddles), 1, 0
#pragma unroll
    for (int r = 0; r < 8; ++r) gfly<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int r = 0; r < 4; ++r) gfly<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
        gfly_triv<MODE, false>(re[g], im[g], re[g + 2], im[g + 2], a.st[1]);
        gfly_triv<MODE, true>(re[g + 1], im[g + 1], re[g + 3], im[g + 3], a.st[1]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly_triv<MODE, false>(re[g], im[g], re[g + 1], im[g + 1], a.st[0]);
    // natural order: X index = brev_L(n) = rev4(r) << (L-4) | n4 << (L-5) | brev(mid) << 8 | rev8(top 8 bits)
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    const size_t off = (frame << L) + ((size_t)(tid >> 8) << (L - 5)) + ((size_t)rmid << 8) + (tid & 255);
    if (a.out16) {
        u32 *dst = static_cast<u32 *>(out) + off;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_nontemporal_store(((u32)re[r] & 0xFFFFu) | ((u32)im[r] << 16), dst + ((size_t)rev4g(r) << (L - 4)));
    } else {
        typedef int v2i __attribute__((ext_vector_type(2)));
        v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + off);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const v2i y = {re[r], im[r]};
            __builtin_nontemporal_store(y, dst + ((size_t)rev4g(r) << (L - 4)));
        }
    }
}

// ---- two-pass split (forward): 2^16 = 256 x 256, as in intfft_big20.hip (k_big20_p1<., ., 8> + k_mid_p2) ---------------
// pass A  STAGE L-1..8 on virtual 2^16-point frames: tile = 256 rows n15..8 (stride 256) x 32 consecutive n7..0, 512 threads,
//         regs = n15..12 -> LDS transpose (one plane, re then im) -> regs = n11..8; both twiddle sets are re-read per
//         tile (L1 / L2 hits) so that the kernel fits 128 VGPRs = two workgroups per CU
// pass B  STAGE 7..0 + bit-reversed store: tile = 32 rows n(L-1)..n(L-5) x 256 consecutive n7..0; thread = (R, n3..0), regs =
//         n7..4 (128-B runs) -> LDS -> thread = (n7..4, rev5(R)), regs = n3..0; the rows are the 5 lowest output bits
// The cores' own beat orders (round 5, NAT instantiations; W32Args::native bit 0: HALVES on the time side, bit 1: BITREV on the frequency side).  Time side:
// register j holds the rows 16 j + hx, so a HALVES beat (x[i], x[i + N/2]) is the register pair (j, j | 2^(L-13)): adjacent samples, one 8- / 16-byte access.
// Frequency side: BITREV order is the core position p = (R << (L-5)) + 256 mid + 16 n7..4 + n3..0; the round layout (thread 32 n7..4 + rev5(R), registers n3..0)
// exchanges its registers with the four low thread bits through the plane, after which a wave instruction covers 32 consecutive positions of two rows.
template <int L> __device__ __forceinline__ constexpr int bw_pair_bit() { return 1 << (L - 13); }
template <int L> __device__ __forceinline__ constexpr int bw_pair_index(int j) { return ((j >> (L - 12)) << (L - 1)) + 4096 * (j & ((1 << (L - 12)) - 1)); }

template <int L, int MODE, bool MASKED, bool NAT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_bigw_a(const void *in, int2 *scr, const int2 *__restrict__ twt, const W32Args a,
                                                size_t nframes_user, unsigned groups)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    constexpr int NS1 = L - 12, G = 1 << (16 - L);
    __shared__ u32 lds[PLANEG3];
    const size_t nframes = (nframes_user + G - 1) / G;
    const int tid = threadIdx.x, l = tid & 31, hx = tid >> 5; // hx = n11..8 (round 1) / n15..12 (round 2)
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups; // chunk 0..7
    const unsigned lfull = chunk * 32 + l;                                   // n7..0
    const unsigned tb1 = (unsigned)hx * 256u + lfull; // round 1: twiddle index = ((j mod 2^i) * 16 + hx) * 256 + lfull
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const bool partial = L < 16 && (frame + 1) * G > nframes_user; // last group: absent frames read as 0, not stored
        unsigned lf = lfull, t1 = tb1; // opaque per tile: keeps the (loop-invariant) twiddle loads inside the loop
        // every global access below is (wave-uniform pointer)[32-bit thread offset] (at32, intfft_device.hpp): 64-bit per-access address pairs cost this
        // kernel 10-18 spilled VGPRs (round 4)
        unsigned toff = (unsigned)hx * 256u + (unsigned)l; // row hx of the register's 16-row block, column l of the chunk
        asm volatile("" : "+v"(lf), "+v"(t1), "+v"(toff));
        int re[16], im[16];
        if (NAT && (a.native & 1)) { // HALVES order in: one access per register pair (j, j | 2^(L-13))
            if (a.in16) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                const v2u *src2 = reinterpret_cast<const v2u *>(in) + frame * 32768 + chunk * 32;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j & bw_pair_bit<L>()) continue;
                    v2u x = {0u, 0u};
                    if (!partial || frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user) x = INTFFT_LD(at32(src2 + bw_pair_index<L>(j), toff));
                    re[j] = (int)(x.x << a.in_sh) >> a.in_sh, im[j] = (int)(x.x << (a.in_sh - 16)) >> a.in_sh;
                    re[j | bw_pair_bit<L>()] = (int)(x.y << a.in_sh) >> a.in_sh, im[j | bw_pair_bit<L>()] = (int)(x.y << (a.in_sh - 16)) >> a.in_sh;
                }
            } else {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i *src4 = reinterpret_cast<const v4i *>(in) + frame * 32768 + chunk * 32;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j & bw_pair_bit<L>()) continue;
                    v4i x = {0, 0, 0, 0};
                    if (!partial || frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user) x = INTFFT_LD(at32(src4 + bw_pair_index<L>(j), toff));
                    re[j] = (int)((u32)x.x << a.in_sh) >> a.in_sh, im[j] = (int)((u32)x.y << a.in_sh) >> a.in_sh;
                    re[j | bw_pair_bit<L>()] = (int)((u32)x.z << a.in_sh) >> a.in_sh, im[j | bw_pair_bit<L>()] = (int)((u32)x.w << a.in_sh) >> a.in_sh;
                }
            }
        } else if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + frame * 65536 + chunk * 32; // wave-uniform
            u32 raw[16];
            if (!partial) { // (one test around the 16 loads: tested one by one they are issued one by one)
#pragma unroll
                for (int j = 0; j < 16; ++j) raw[j] = INTFFT_LD(at32(src + ((size_t)j << 12), toff));
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    raw[j] = frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user ? INTFFT_LD(at32(src + ((size_t)j << 12), toff)) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                re[j] = (int)(raw[j] << a.in_sh) >> a.in_sh, im[j] = (int)(raw[j] << (a.in_sh - 16)) >> a.in_sh;
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + frame * 65536 + chunk * 32);
            v2i x[16];
            if (!partial) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = INTFFT_LD(at32(src + ((size_t)j << 12), toff));
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x[j] = v2i{0, 0};
                    if (frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user) x[j] = INTFFT_LD(at32(src + ((size_t)j << 12), toff));
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                re[j] = (int)((u32)x[j].x << a.in_sh) >> a.in_sh, im[j] = (int)((u32)x[j].y << a.in_sh) >> a.in_sh;
        }
        {
            int w8r[8] = {}, w8i[8] = {}, w4r[4] = {}, w4i[4] = {}, w2r[2] = {}, w2i[2] = {}, w1r, w1i;
            int2 w;
            if constexpr (NS1 >= 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w = twt[(1u << 15) - 1u + t1 + (unsigned)j * 4096u], w8r[j] = w.x, w8i[j] = w.y;
            }
            if constexpr (NS1 >= 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w = twt[(1u << 14) - 1u + t1 + (unsigned)j * 4096u], w4r[j] = w.x, w4i[j] = w.y;
            }
            if constexpr (NS1 >= 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) w = twt[(1u << 13) - 1u + t1 + (unsigned)j * 4096u], w2r[j] = w.x, w2i[j] = w.y;
            }
            w = twt[(1u << 12) - 1u + t1], w1r = w.x, w1i = w.y;
            gstages<MODE, MASKED, NS1, 12>(re, im, w8r, w8i, w4r, w4i, w2r, w2i, w1r, w1i, a);
        }
        // STAGE 11..8 twiddles (index = (reg mod 2^i) * 256 + lfull): issued here, they arrive during the transpose
        int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
        {
            int2 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w = twt[2047u + lf + 256u * j], b8r[j] = w.x, b8i[j] = w.y;
#pragma unroll
            for (int j = 0; j < 4; ++j) w = twt[1023u + lf + 256u * j], b4r[j] = w.x, b4i[j] = w.y;
#pragma unroll
            for (int j = 0; j < 2; ++j) w = twt[511u + lf + 256u * j], b2r[j] = w.x, b2i[j] = w.y;
            w = twt[255u + lf], b1r = w.x, b1i = w.y;
        }
        // transpose: (thread (hx = n11..8, l), reg j = n15..12) -> (thread (j, l), reg hx); one plane, re then im
        {
            u32 *w = lds + ROWG * l + hx;
#pragma unroll
            for (int j = 0; j < 16; ++j) w[ROWG * 32 * j] = (u32)re[j];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) re[r] = (int)lds[ROWG * tid + r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) w[ROWG * 32 * j] = (u32)im[j];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) im[r] = (int)lds[ROWG * tid + r];
            __syncthreads(); // the next tile's writes stay behind these reads
        }
        gstages<MODE, MASKED, 4, 8>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
        {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i *dst = reinterpret_cast<v2i *>(scr + frame * 65536 + chunk * 32); // wave-uniform
            const unsigned soff = (unsigned)hx * 4096u + (toff & 31u);            // row 16 hx + r: r goes into the uniform part
            if (!partial) {
#pragma unroll
                for (int r = 0; r < 16; ++r) *at32(dst + ((size_t)r << 8), soff) = v2i{re[r], im[r]};
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (frame * G + (size_t)((16 * hx + r) >> (L - 8)) < nframes_user) *at32(dst + ((size_t)r << 8), soff) = v2i{re[r], im[r]};
            }
        }
    }
}

template <int MODE, bool MASKED, bool NAT = false>
__global__ __launch_bounds__(512) void k_bigw_b(const int2 *scr, void *out, const int2 *__restrict__ twt, const UConsts c,
                                                const W32Args a, int L)
{
    __shared__ u32 lds[PLANEG3];
    const int tid = threadIdx.x, lo4 = tid & 15, R = tid >> 4;
    const size_t frame = blockIdx.x >> (L - 13);
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u)); // n(L-6)..n8
    int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
    {
        int2 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w = twt[127 + 16 * j + lo4], b8r[j] = w.x, b8i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) w = twt[63 + 16 * j + lo4], b4r[j] = w.x, b4i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 2; ++j) w = twt[31 + 16 * j + lo4], b2r[j] = w.x, b2i[j] = w.y;
        w = twt[15 + lo4], b1r = w.x, b1i = w.y;
    }
    const int2 *src = scr + (frame << L) + ((size_t)R << (L - 5)) + mid * 256 + lo4;
    int re[16], im[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int2 x = src[16 * j]; // regs = n7..4
        re[j] = x.x, im[j] = x.y;
    }
    gstages<MODE, MASKED, 4, 4>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
    // transpose: (thread (R, n3..0), reg j = n7..4) -> (thread 32 j + rev5(R), reg n3..0)
    {
        u32 *w = lds + ROWG * (int)(__brev((unsigned)R) >> 27) + lo4;
#pragma unroll
        for (int j = 0; j < 16; ++j) w[ROWG * 32 * j] = (u32)re[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)lds[ROWG * tid + r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) w[ROWG * 32 * j] = (u32)im[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) im[r] = (int)lds[ROWG * tid + r];
    }
    // stages 3, 2 (uniform twiddles), 1, 0
#pragma unroll
    for (int r = 0; r < 8; ++r) gfly<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int r = 0; r < 4; ++r) gfly<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
        gfly_triv<MODE, false>(re[g], im[g], re[g + 2], im[g + 2], a.st[1]);
        gfly_triv<MODE, true>(re[g + 1], im[g + 1], re[g + 3], im[g + 3], a.st[1]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly_triv<MODE, false>(re[g], im[g], re[g + 1], im[g + 1], a.st[0]);
    if (NAT && (a.native & 2)) { // BITREV order out: memory index = core position
        // registers n3..0 <-> the four low thread bits (rev5(R) & 15), one plane, re then im: thread = (n7..4, R bit 0, n3..0), register x = rev5(R) & 15
        const u32 *xr = lds + ROWG * (tid & ~15) + (tid & 15); // + ROWG * x
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[ROWG * tid + r] = (u32)re[r];
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 16; ++x) re[x] = (int)xr[ROWG * x];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[ROWG * tid + r] = (u32)im[r];
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 16; ++x) im[x] = (int)xr[ROWG * x];
        // R = rev5(x | b << 4) = rev4(x) << 1 | b, b = thread bit 4: position = (R << (L-5)) + 256 mid + 16 (tid >> 5) + (tid & 15)
        const size_t offp = (frame << L) + ((size_t)((tid >> 4) & 1) << (L - 5)) + mid * 256 + 16 * (tid >> 5) + (tid & 15);
        if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + offp;
#pragma unroll
            for (int x = 0; x < 16; ++x) __builtin_nontemporal_store(((u32)re[x] & 0xFFFFu) | ((u32)im[x] << 16), dst + ((size_t)rev4g(x) << (L - 4)));
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + offp);
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const v2i y = {re[x], im[x]};
                __builtin_nontemporal_store(y, dst + ((size_t)rev4g(x) << (L - 4)));
            }
        }
        return;
    }
    // natural order: X index = brev_L(n) = rev4(n3..0) << (L-4) | rev4(n7..4) << (L-8) | brev(mid) << 5 | rev5(R)
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    const size_t off = (frame << L) + ((size_t)rev4g(tid >> 5) << (L - 8)) + ((size_t)rmid << 5) + (tid & 31);
    if (a.out16) {
        u32 *dst = static_cast<u32 *>(out) + off;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_nontemporal_store(((u32)re[r] & 0xFFFFu) | ((u32)im[r] << 16), dst + ((size_t)rev4g(r) << (L - 4)));
    } else {
        typedef int v2i __attribute__((ext_vector_type(2)));
        v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + off);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const v2i y = {re[r], im[r]};
            __builtin_nontemporal_store(y, dst + ((size_t)rev4g(r) << (L - 4)));
        }
    }
}

// ---- inverse passes (int_ifftNk, mirrors of the three passes above) ------------------------------------------------------
// four DIT stages s0 .. s0+3 on register offsets 1, 2, 4, 8; only the first NS of them
template <int MODE, bool MASKED, int NS, int S0>
__device__ __forceinline__ void gstages_dit(int (&re)[16], int (&im)[16], const int (&w8r)[8], const int (&w8i)[8],
                                            const int (&w4r)[4], const int (&w4i)[4], const int (&w2r)[2], const int (&w2i)[2],
                                            int w1r, int w1i, const W32Args &a)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly_dit<MODE, false, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w1r, w1i, a.st[S0]);
    if constexpr (NS >= 2) {
#pragma unroll
        for (int g = 0; g < 16; g += 4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w2r[j], w2i[j], a.st[S0 + 1]);
    }
    if constexpr (NS >= 3) {
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gfly_dit<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r[j], w4i[j], a.st[S0 + 2]);
    }
    if constexpr (NS >= 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gfly_dit<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w8r[j], w8i[j], a.st[S0 + 3]);
    }
}

// inverse pass 3: bit-reversed load of the natural-order input (1-2 KiB runs) + DIT 0..3, user array -> scratch
template <int MODE, bool MASKED>
__global__ __launch_bounds__(512) void k_bigw_q3(const void *in, int2 *scr, const UConsts c, const W32Args a, int L)
{
    __shared__ u32 lds[PLANEG3];
    const int tid = threadIdx.x;
    const size_t frame = blockIdx.x >> (L - 13);
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u));
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    const size_t off = (frame << L) + ((size_t)(tid >> 8) << (L - 5)) + ((size_t)rmid << 8) + (tid & 255);
    int re[16], im[16];
    if (a.in16) {
        const u32 *src = static_cast<const u32 *>(in) + off;
        u32 raw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[r] = INTFFT_LD(src + ((size_t)rev4g(r) << (L - 4)));
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)(raw[r] << a.in_sh) >> a.in_sh, im[r] = (int)(raw[r] << (a.in_sh - 16)) >> a.in_sh;
    } else {
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + off);
        v2i x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = INTFFT_LD(src + ((size_t)rev4g(r) << (L - 4)));
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)((u32)x[r].x << a.in_sh) >> a.in_sh, im[r] = (int)((u32)x[r].y << a.in_sh) >> a.in_sh;
    }
    // DIT 0..3 on regs n3..0
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
        for (int r = 0; r < 4; ++r) gfly_dit<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
#pragma unroll
    for (int r = 0; r < 8; ++r) gfly_dit<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
    // transpose to thread = (px = n(L-5)..n(L-8), e = n4..0), regs q = n(L-1)..n(L-4): rev8(16 q + px) = tid & 255
    const int px = rev4g((tid >> 4) & 15), jq = rev4g(tid & 15);
    u32 *w = lds + ROWG * (32 * px + 16 * (tid >> 8)) + jq;
#pragma unroll
    for (int r = 0; r < 16; ++r) w[ROWG * r] = (u32)re[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) re[q] = (int)lds[ROWG * tid + q];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) w[ROWG * r] = (u32)im[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) im[q] = (int)lds[ROWG * tid + q];
    int2 *dst = scr + (frame << L) + mid * 32 + (tid & 31);
#pragma unroll
    for (int q = 0; q < 16; ++q) dst[(size_t)(16 * q + (tid >> 5)) << (L - 8)] = make_int2(re[q], im[q]);
}

// inverse pass 2: DIT 4..11 on every 4096-point block, in place
template <int MODE, bool MASKED>
__global__ __launch_bounds__(256) void k_bigw_q2(int2 *scr, const int2 *__restrict__ twt, const W32Args a, size_t nblocks4k)
{
    __shared__ u32 lds[2 * PLANEG2];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    int a8r[8], a8i[8], a4r[4], a4i[4], a2r[2], a2i[2], a1r, a1i;
    int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
    {
        int2 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w = twt[2047 + 256 * j + tid], a8r[j] = w.x, a8i[j] = w.y;
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
    u32 *const wr = lds + ROWG * lo4 + hi4;
    const u32 *const rd = lds + ROWG * tid;
    auto transpose = [&](int (&re)[16], int (&im)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            wr[ROWG * 16 * j] = (u32)re[j];
            wr[PLANEG2 + ROWG * 16 * j] = (u32)im[j];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)rd[r], im[r] = (int)rd[PLANEG2 + r];
        __syncthreads();
    };
    for (size_t b = blockIdx.x; b < nblocks4k; b += gridDim.x) {
        int2 *p = scr + b * 4096;
        int re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int2 x = p[256 * j + tid]; // LA
            re[j] = x.x, im[j] = x.y;
        }
        transpose(re, im); // LB: regs = n7..4
        gstages_dit<MODE, MASKED, 4, 4>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
        transpose(re, im); // LA: regs = n11..8
        gstages_dit<MODE, MASKED, 4, 8>(re, im, a8r, a8i, a4r, a4i, a2r, a2i, a1r, a1i, a);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[256 * r + tid] = make_int2(re[r], im[r]);
    }
}

// inverse pass 1: DIT 12..L-1 on groups of 2^(16-L) frames, scratch -> user array (natural order)
template <int L, int MODE, bool MASKED>
__global__ __launch_bounds__(512) void k_bigw_q1(const int2 *scr, void *out, const int2 *__restrict__ twt, const W32Args a,
                                                 size_t nframes_user, unsigned groups)
{
    static_assert(L >= 13 && L <= 16, "one-round pass");
    constexpr int NS = L - 12, G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G;
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups;
    const unsigned lfull = chunk * 512 + threadIdx.x;
    int w8r[8] = {}, w8i[8] = {}, w4r[4] = {}, w4i[4] = {}, w2r[2] = {}, w2i[2] = {}, w1r, w1i;
    {
        int2 w;
        if constexpr (NS >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w = twt[(1u << 15) - 1u + lfull + (unsigned)j * 4096u], w8r[j] = w.x, w8i[j] = w.y;
        }
        if constexpr (NS >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w = twt[(1u << 14) - 1u + lfull + (unsigned)j * 4096u], w4r[j] = w.x, w4i[j] = w.y;
        }
        if constexpr (NS >= 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) w = twt[(1u << 13) - 1u + lfull + (unsigned)j * 4096u], w2r[j] = w.x, w2i[j] = w.y;
        }
        w = twt[(1u << 12) - 1u + lfull], w1r = w.x, w1i = w.y;
    }
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const bool partial = L < 16 && (frame + 1) * G > nframes_user;
        const int2 *src = scr + frame * 65536 + lfull;
        int re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int2 x = make_int2(0, 0);
            if (!partial || frame * G + (size_t)(j >> (L - 12)) < nframes_user) x = src[(size_t)j << 12];
            re[j] = x.x, im[j] = x.y;
        }
        gstages_dit<MODE, MASKED, NS, 12>(re, im, w8r, w8i, w4r, w4i, w2r, w2i, w1r, w1i, a);
        if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + frame * 65536 + lfull;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || frame * G + (size_t)(j >> (L - 12)) < nframes_user)
                    __builtin_nontemporal_store(((u32)re[j] & 0xFFFFu) | ((u32)im[j] << 16), dst + ((size_t)j << 12));
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + frame * 65536 + lfull);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || frame * G + (size_t)(j >> (L - 12)) < nframes_user) {
                    const v2i y = {re[j], im[j]};
                    __builtin_nontemporal_store(y, dst + ((size_t)j << 12));
                }
        }
    }
}

// ---- inverse two-pass split (mirrors of k_bigw_b / k_bigw_a) ----------------------------------------------------------------
// first pass: bit-reversed load of the natural-order input (128-B runs), thread = (n7..4, rev5(R)), regs = n3..0: DIT 0..3,
// LDS transpose to thread = (R, n3..0), regs = n7..4: DIT 4..7, scratch
template <int MODE, bool MASKED, bool NAT = false>
__global__ __launch_bounds__(512) void k_bigw_qb(const void *in, int2 *scr, const int2 *__restrict__ twt, const UConsts c,
                                                 const W32Args a, int L)
{
    __shared__ u32 lds[PLANEG3];
    const int tid = threadIdx.x, lo4 = tid & 15, R = tid >> 4;
    const size_t frame = blockIdx.x >> (L - 13);
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u)); // n(L-6)..n8
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    const size_t off = (frame << L) + ((size_t)rev4g(tid >> 5) << (L - 8)) + ((size_t)rmid << 5) + (tid & 31);
    int re[16], im[16];
    if (NAT && (a.native & 2)) { // BITREV order in: memory index = core position; coalesced loads (thread = (n7..4, R bit 0, n3..0), register x = rev5(R) & 15),
        // then the exchange of k_bigw_b run backwards
        const size_t offp = (frame << L) + ((size_t)((tid >> 4) & 1) << (L - 5)) + mid * 256 + 16 * (tid >> 5) + (tid & 15);
        if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + offp;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const u32 raw = INTFFT_LD(src + ((size_t)rev4g(x) << (L - 4)));
                re[x] = (int)(raw << a.in_sh) >> a.in_sh, im[x] = (int)(raw << (a.in_sh - 16)) >> a.in_sh;
            }
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + offp);
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const v2i v = INTFFT_LD(src + ((size_t)rev4g(x) << (L - 4)));
                re[x] = (int)((u32)v.x << a.in_sh) >> a.in_sh, im[x] = (int)((u32)v.y << a.in_sh) >> a.in_sh;
            }
        }
        u32 *xw = lds + ROWG * (tid & ~15) + (tid & 15); // + ROWG * x
#pragma unroll
        for (int x = 0; x < 16; ++x) xw[ROWG * x] = (u32)re[x];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)lds[ROWG * tid + r];
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 16; ++x) xw[ROWG * x] = (u32)im[x];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) im[r] = (int)lds[ROWG * tid + r];
        __syncthreads();
    } else if (a.in16) {
        const u32 *src = static_cast<const u32 *>(in) + off;
        u32 raw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[r] = INTFFT_LD(src + ((size_t)rev4g(r) << (L - 4)));
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)(raw[r] << a.in_sh) >> a.in_sh, im[r] = (int)(raw[r] << (a.in_sh - 16)) >> a.in_sh;
    } else {
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i *src = reinterpret_cast<const v2i *>(static_cast<const int2 *>(in) + off);
        v2i x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = INTFFT_LD(src + ((size_t)rev4g(r) << (L - 4)));
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = (int)((u32)x[r].x << a.in_sh) >> a.in_sh, im[r] = (int)((u32)x[r].y << a.in_sh) >> a.in_sh;
    }
    // DIT 0..3 on regs n3..0
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
        for (int r = 0; r < 4; ++r) gfly_dit<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
#pragma unroll
    for (int r = 0; r < 8; ++r) gfly_dit<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
    int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
    {
        int2 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w = twt[127 + 16 * j + lo4], b8r[j] = w.x, b8i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) w = twt[63 + 16 * j + lo4], b4r[j] = w.x, b4i[j] = w.y;
#pragma unroll
        for (int j = 0; j < 2; ++j) w = twt[31 + 16 * j + lo4], b2r[j] = w.x, b2i[j] = w.y;
        w = twt[15 + lo4], b1r = w.x, b1i = w.y;
    }
    // transpose: (thread 32 hi4 + rev5(R), reg n3..0) -> (thread (R, n3..0), reg j = n7..4)
    {
        const u32 *rd = lds + ROWG * (int)(__brev((unsigned)R) >> 27) + lo4;
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[ROWG * tid + r] = (u32)re[r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) re[j] = (int)rd[ROWG * 32 * j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[ROWG * tid + r] = (u32)im[r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) im[j] = (int)rd[ROWG * 32 * j];
    }
    gstages_dit<MODE, MASKED, 4, 4>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
    int2 *dst = scr + (frame << L) + ((size_t)R << (L - 5)) + mid * 256 + lo4;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[16 * j] = make_int2(re[j], im[j]);
}

// second pass: DIT 8..L-1 on virtual 2^16-point frames (tiles, twiddle re-reads and occupancy as k_bigw_a), scratch -> user
template <int L, int MODE, bool MASKED, bool NAT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_bigw_qa(const int2 *scr, void *out, const int2 *__restrict__ twt, const W32Args a,
                                                 size_t nframes_user, unsigned groups)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    constexpr int NS1 = L - 12, G = 1 << (16 - L);
    __shared__ u32 lds[PLANEG3];
    const size_t nframes = (nframes_user + G - 1) / G;
    const int tid = threadIdx.x, l = tid & 31, hx = tid >> 5; // hx = n15..12 (round 1: regs n11..8) / n11..8 (round 2: regs n15..12)
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups;
    const unsigned lfull = chunk * 32 + l;
    const unsigned tb1 = (unsigned)hx * 256u + lfull;
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const bool partial = L < 16 && (frame + 1) * G > nframes_user;
        unsigned lf = lfull, t1 = tb1; // opaque per tile: keeps the twiddle loads inside the loop (see k_bigw_a)
        asm volatile("" : "+v"(lf), "+v"(t1));
        const int2 *src = scr + frame * 65536 + lfull;
        int re[16], im[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int2 x = make_int2(0, 0);
            if (!partial || frame * G + (size_t)((16 * hx + r) >> (L - 8)) < nframes_user) x = src[(size_t)(16 * hx + r) << 8];
            re[r] = x.x, im[r] = x.y;
        }
        {
            int b8r[8], b8i[8], b4r[4], b4i[4], b2r[2], b2i[2], b1r, b1i;
            int2 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w = twt[2047u + lf + 256u * j], b8r[j] = w.x, b8i[j] = w.y;
#pragma unroll
            for (int j = 0; j < 4; ++j) w = twt[1023u + lf + 256u * j], b4r[j] = w.x, b4i[j] = w.y;
#pragma unroll
            for (int j = 0; j < 2; ++j) w = twt[511u + lf + 256u * j], b2r[j] = w.x, b2i[j] = w.y;
            w = twt[255u + lf], b1r = w.x, b1i = w.y;
            gstages_dit<MODE, MASKED, 4, 8>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
        }
        int w8r[8] = {}, w8i[8] = {}, w4r[4] = {}, w4i[4] = {}, w2r[2] = {}, w2i[2] = {}, w1r, w1i;
        {
            int2 w; // STAGE 12..15 twiddles for the thread after the transpose: its tid >> 5 is then n11..8, as tb1 assumes
            if constexpr (NS1 >= 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w = twt[(1u << 15) - 1u + t1 + (unsigned)j * 4096u], w8r[j] = w.x, w8i[j] = w.y;
            }
            if constexpr (NS1 >= 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w = twt[(1u << 14) - 1u + t1 + (unsigned)j * 4096u], w4r[j] = w.x, w4i[j] = w.y;
            }
            if constexpr (NS1 >= 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) w = twt[(1u << 13) - 1u + t1 + (unsigned)j * 4096u], w2r[j] = w.x, w2i[j] = w.y;
            }
            w = twt[(1u << 12) - 1u + t1], w1r = w.x, w1i = w.y;
        }
        // transpose: (thread (hx = n15..12, l), reg r = n11..8) -> (thread (r, l), reg hx)
        {
            u32 *w = lds + ROWG * l + hx;
#pragma unroll
            for (int r = 0; r < 16; ++r) w[ROWG * 32 * r] = (u32)re[r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) re[j] = (int)lds[ROWG * tid + j];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) w[ROWG * 32 * r] = (u32)im[r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) im[j] = (int)lds[ROWG * tid + j];
            __syncthreads();
        }
        gstages_dit<MODE, MASKED, NS1, 12>(re, im, w8r, w8i, w4r, w4i, w2r, w2i, w1r, w1i, a); // regs j = n15..12, thread hx = n11..8
        if (NAT && (a.native & 1)) { // HALVES order out: one store per register pair (j, j | 2^(L-13))
            const unsigned toffp = (unsigned)hx * 256u + lfull;
            if (a.out16) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                v2u *dst2 = reinterpret_cast<v2u *>(out) + frame * 32768;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j & bw_pair_bit<L>()) continue;
                    constexpr int PB = bw_pair_bit<L>();
                    const v2u y = {((u32)re[j] & 0xFFFFu) | ((u32)im[j] << 16), ((u32)re[j | PB] & 0xFFFFu) | ((u32)im[j | PB] << 16)};
                    if (!partial || frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user) __builtin_nontemporal_store(y, dst2 + bw_pair_index<L>(j) + toffp);
                }
            } else {
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i *dst4 = reinterpret_cast<v4i *>(out) + frame * 32768;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j & bw_pair_bit<L>()) continue;
                    constexpr int PB = bw_pair_bit<L>();
                    const v4i y = {re[j], im[j], re[j | PB], im[j | PB]};
                    if (!partial || frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user) __builtin_nontemporal_store(y, dst4 + bw_pair_index<L>(j) + toffp);
                }
            }
            continue;
        }
        if (a.out16) {
            u32 *dst = static_cast<u32 *>(out) + frame * 65536 + lfull;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user)
                    __builtin_nontemporal_store(((u32)re[j] & 0xFFFFu) | ((u32)im[j] << 16), dst + ((size_t)(16 * j + hx) << 8));
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + frame * 65536 + lfull);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!partial || frame * G + (size_t)((16 * j + hx) >> (L - 8)) < nframes_user) {
                    const v2i y = {re[j], im[j]};
                    __builtin_nontemporal_store(y, dst + ((size_t)(16 * j + hx) << 8));
                }
        }
    }
}

bool bigw_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                    int out_order)
{
    return log2n >= 13 && log2n <= 16 && data_width >= 2 && data_width + format * log2n <= 32 && twdl_width >= 4 &&
           twdl_width <= 26 && (direction == 0 || direction == 1) && use_fly == 1 &&
           (direction == 0 ? (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1)  // int_fftNk: NATURAL | HALVES in, NATURAL | BITREV out
                           : (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2)); // int_ifftNk: NATURAL | BITREV in, NATURAL | HALVES out
    // (the cores' own orders: on the two-pass kernels only -- under INTFFT_NO_TWOPASS the planner keeps the generic passes for them)
}

const char *bigw_kernel_name(int direction, int two_pass)
{
    if (two_pass == 2) return direction == 1 ? "k_bigw_qb/qa+k_bigw_post" : "k_bigw_pre+k_bigw_a/b"; // N = 2^17 .. 2^20
    return direction == 1 ? (two_pass ? "k_bigw_qb/qa" : "k_bigw_q3/q2/q1") : two_pass ? "k_bigw_a/b" : "k_bigw_p1/p2/p3";
}

template <int MODE, bool MASKED>
static hipError_t launch_bigw_inv(int log2n, const W32Args &a, const void *in, void *out, int2 *scr, const int2 *tw, const UConsts &c,
                                  size_t nframes, hipStream_t stream)
{
    const size_t nb = nframes << (log2n - 12), cap = resident_blocks(kptr(k_bigw_q2<MODE, MASKED>), 256, 2, 0, false), nb3 = nframes << (log2n - 13);
    if (nb3 > 0x7fffffffull) return hipErrorInvalidValue;
    if (log2n > 16) { // N = 2^17 .. 2^20 (round 5, intfft_bigwlong.hip): the gather pass at L = NFFT, STAGE 8 .. 15 in place on the 2^16-point blocks, STAGE 16 .. in a post-pass
        W32Args a1 = a;
        a1.out16 = 0; // the scratch holds int32 pairs
        const size_t nblocks = nframes << (log2n - 16);
        const unsigned ga = (unsigned)(nblocks < 256 ? nblocks : 256);
        a1.native = 0; // the blocks in place: plain core positions (BITREV in: k_bigw_qb<NAT> at L = NFFT; HALVES out: k_bigw_post)
        if (a.native & 2) hipLaunchKernelGGL((k_bigw_qb<MODE, MASKED, true>), dim3((unsigned)nb3), dim3(512), 0, stream, in, scr, tw, c, a, log2n);
        else hipLaunchKernelGGL((k_bigw_qb<MODE, MASKED>), dim3((unsigned)nb3), dim3(512), 0, stream, in, scr, tw, c, a, log2n);
        hipLaunchKernelGGL((k_bigw_qa<16, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, scr, scr, tw, a1, nblocks, ga);
        return launch_bigw_post(log2n, MODE, a, scr, out, tw, nframes, stream);
    }
    if (a.two_pass) {
        const size_t nvfa = (nframes + ((size_t)1 << (16 - log2n)) - 1) >> (16 - log2n);
        const unsigned ga = (unsigned)(nvfa < 256 ? nvfa : 256);
        if (a.native) { // the natural-order instantiations carry none of the native-order code
            hipLaunchKernelGGL((k_bigw_qb<MODE, MASKED, true>), dim3((unsigned)nb3), dim3(512), 0, stream, in, scr, tw, c, a, log2n);
            switch (log2n) {
            case 13: hipLaunchKernelGGL((k_bigw_qa<13, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
            case 14: hipLaunchKernelGGL((k_bigw_qa<14, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
            case 15: hipLaunchKernelGGL((k_bigw_qa<15, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
            default: hipLaunchKernelGGL((k_bigw_qa<16, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
            }
            return hipGetLastError();
        }
        hipLaunchKernelGGL((k_bigw_qb<MODE, MASKED>), dim3((unsigned)nb3), dim3(512), 0, stream, in, scr, tw, c, a, log2n);
        switch (log2n) {
        case 13: hipLaunchKernelGGL((k_bigw_qa<13, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
        case 14: hipLaunchKernelGGL((k_bigw_qa<14, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
        case 15: hipLaunchKernelGGL((k_bigw_qa<15, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
        default: hipLaunchKernelGGL((k_bigw_qa<16, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, scr, out, tw, a, nframes, ga); break;
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_bigw_q3<MODE, MASKED>), dim3((unsigned)nb3), dim3(512), 0, stream, in, scr, c, a, log2n);
    hipLaunchKernelGGL((k_bigw_q2<MODE, MASKED>), dim3((unsigned)(nb < cap ? nb : cap)), dim3(256), 0, stream, scr, tw, a, nb);
    const size_t nvf = (nframes + ((size_t)1 << (16 - log2n)) - 1) >> (16 - log2n);
    const unsigned groups = (unsigned)(nvf < 128 ? nvf : 128);
    switch (log2n) {
    case 13: hipLaunchKernelGGL((k_bigw_q1<13, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, scr, out, tw, a, nframes, groups); break;
    case 14: hipLaunchKernelGGL((k_bigw_q1<14, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, scr, out, tw, a, nframes, groups); break;
    case 15: hipLaunchKernelGGL((k_bigw_q1<15, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, scr, out, tw, a, nframes, groups); break;
    default: hipLaunchKernelGGL((k_bigw_q1<16, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, scr, out, tw, a, nframes, groups); break;
    }
    return hipGetLastError();
}

template <int MODE, bool MASKED>
static hipError_t launch_bigw_m(int log2n, const W32Args &a, const void *in, void *out, int2 *scr, const int2 *tw, const UConsts &c,
                                size_t nframes, hipStream_t stream)
{
    if (a.inverse) return launch_bigw_inv<MODE, MASKED>(log2n, a, in, out, scr, tw, c, nframes, stream);
    if (log2n > 16) { // N = 2^17 .. 2^20 (round 5, intfft_bigwlong.hip): STAGE NFFT-1 .. 16 in a pre-pass, then the two passes -- pass A in place on the 2^16-point blocks
        const hipError_t e = launch_bigw_pre(log2n, MODE, a, in, scr, tw, nframes, stream);
        if (e != hipSuccess) return e;
        W32Args a1 = a;
        a1.in16 = 0, a1.in_sh = 0, a1.native = 0; // the scratch holds wrapped int32 pairs at plain core positions (HALVES in: k_bigw_pre; BITREV out: k_bigw_b<NAT>)
        const size_t nblocks = nframes << (log2n - 16), nb2 = nframes << (log2n - 13);
        if (nb2 > 0x7fffffffull) return hipErrorInvalidValue;
        const unsigned ga = (unsigned)(nblocks < 256 ? nblocks : 256);
        hipLaunchKernelGGL((k_bigw_a<16, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, scr, scr, tw, a1, nblocks, ga);
        if (a.native & 2) hipLaunchKernelGGL((k_bigw_b<MODE, MASKED, true>), dim3((unsigned)nb2), dim3(512), 0, stream, scr, out, tw, c, a, log2n);
        else hipLaunchKernelGGL((k_bigw_b<MODE, MASKED>), dim3((unsigned)nb2), dim3(512), 0, stream, scr, out, tw, c, a, log2n);
        return hipGetLastError();
    }
    const size_t nvf = (nframes + ((size_t)1 << (16 - log2n)) - 1) >> (16 - log2n);
    if (a.two_pass) {
        const unsigned ga = (unsigned)(nvf < 256 ? nvf : 256);
        const size_t nb2 = nframes << (log2n - 13);
        if (nb2 > 0x7fffffffull) return hipErrorInvalidValue;
        if (a.native) {
            switch (log2n) {
            case 13: hipLaunchKernelGGL((k_bigw_a<13, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
            case 14: hipLaunchKernelGGL((k_bigw_a<14, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
            case 15: hipLaunchKernelGGL((k_bigw_a<15, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
            default: hipLaunchKernelGGL((k_bigw_a<16, MODE, MASKED, true>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
            }
            hipLaunchKernelGGL((k_bigw_b<MODE, MASKED, true>), dim3((unsigned)nb2), dim3(512), 0, stream, scr, out, tw, c, a, log2n);
            return hipGetLastError();
        }
        switch (log2n) {
        case 13: hipLaunchKernelGGL((k_bigw_a<13, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
        case 14: hipLaunchKernelGGL((k_bigw_a<14, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
        case 15: hipLaunchKernelGGL((k_bigw_a<15, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
        default: hipLaunchKernelGGL((k_bigw_a<16, MODE, MASKED>), dim3(8u * ga), dim3(512), 0, stream, in, scr, tw, a, nframes, ga); break;
        }
        hipLaunchKernelGGL((k_bigw_b<MODE, MASKED>), dim3((unsigned)nb2), dim3(512), 0, stream, scr, out, tw, c, a, log2n);
        return hipGetLastError();
    }
    const unsigned groups = (unsigned)(nvf < 128 ? nvf : 128);
    switch (log2n) {
    case 13: hipLaunchKernelGGL((k_bigw_p1<13, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, in, scr, tw, a, nframes, groups); break;
    case 14: hipLaunchKernelGGL((k_bigw_p1<14, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, in, scr, tw, a, nframes, groups); break;
    case 15: hipLaunchKernelGGL((k_bigw_p1<15, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, in, scr, tw, a, nframes, groups); break;
    default: hipLaunchKernelGGL((k_bigw_p1<16, MODE, MASKED>), dim3(8u * groups), dim3(512), 0, stream, in, scr, tw, a, nframes, groups); break;
    }
    const size_t nb = nframes << (log2n - 12), cap = resident_blocks(kptr(k_bigw_p2<MODE, MASKED>), 256, 2, 0, false), nb3 = nframes << (log2n - 13);
    if (nb3 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_bigw_p2<MODE, MASKED>), dim3((unsigned)(nb < cap ? nb : cap)), dim3(256), 0, stream, scr, tw, a, nb);
    hipLaunchKernelGGL((k_bigw_p3<MODE, MASKED>), dim3((unsigned)nb3), dim3(512), 0, stream, scr, out, c, a, log2n);
    return hipGetLastError();
}

hipError_t launch_bigw(int log2n, int mode, const W32Args &a, const void *in, void *out, void *scratch, const int2 *tw_all,
                       const int2 *h_tw, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    int2 *scr = static_cast<int2 *>(scratch);
    if (a.masked) {
        switch (mode) {
        case W_TRUNC: return launch_bigw_m<W_TRUNC, true>(log2n, a, in, out, scr, tw_all, c, nframes, stream);
        case W_ROUND: return launch_bigw_m<W_ROUND, true>(log2n, a, in, out, scr, tw_all, c, nframes, stream);
        default: return launch_bigw_m<W_UNSCALED, true>(log2n, a, in, out, scr, tw_all, c, nframes, stream);
        }
    }
    switch (mode) {
    case W_TRUNC: return launch_bigw_m<W_TRUNC, false>(log2n, a, in, out, scr, tw_all, c, nframes, stream);
    case W_ROUND: return launch_bigw_m<W_ROUND, false>(log2n, a, in, out, scr, tw_all, c, nframes, stream);
    default: return launch_bigw_m<W_UNSCALED, false>(log2n, a, in, out, scr, tw_all, c, nframes, stream);
    }
}

} // namespace intfft
