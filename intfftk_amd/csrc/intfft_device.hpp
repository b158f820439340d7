// intfft_device.hpp -- device arithmetic of the fixed-point radix-2 butterflies (gfx950).
//
// One inline device function per arithmetic block of the reference RTL (hukenovs/intfftk):
//   int_addsub_dsp48   src/vhdl/math/int_addsub_dsp48.vhd:17-22      -> plain +/-
//   int_cmult_dsp48    src/vhdl/math/cmult/int_cmult_dsp48.vhd:182-434 -> cmult<T>()
//   int_dif2_fly       src/vhdl/fft/int_dif2_fly.vhd:144-373          -> dif_fly<T>()
//   int_dit2_fly       src/vhdl/fft/int_dit2_fly.vhd:142-325          -> dit_fly<T>()
// T is the on-chip word: int32_t when every width of the plan is <= 32 bits, else int64_t.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// Streaming loads of user / scratch data.  Round 3 measured the four combinations of plain / non-temporal loads and stores on the
// headline kernel (same box, alternating): nt loads + nt stores 704 Gsample/s (rounds 1-2), nt loads + plain stores 728, plain
// loads + plain stores 650, PLAIN loads + nt stores 767-830.  Stores stay non-temporal everywhere.  Loads: the single-pass kernels
// (one read of the user array, one write) gain 3-30 % from plain loads (C2 830 against 739, inverse 803 / 710, N = 4096 forward
// 696 / 631, the C5 pair 401 / 389, 24-bit unscaled N = 1024 268 / 205); the multi-pass kernels LOSE 4-14 % with them (C4 269
// against 291, C3 133 / 144, N = 2^16 313 / 357) and define INTFFT_NT_LOADS at the top of their translation units.
#ifdef INTFFT_NT_LOADS
#define INTFFT_LD(p) __builtin_nontemporal_load(p)
#else
#define INTFFT_LD(p) (*(p))
#endif

namespace intfft {

// (wave-uniform pointer)[per-thread 32-bit element offset] formed as SGPR base + zero-extended 32-bit VGPR BYTE offset: the saddr form of
// global_load / global_store.  Written as p[off] the compiler re-associates (base + thread offset) + constant, keeps the sum in a VGPR
// pair and spends a v_add_co / v_addc pair on every access (k_big2x_a: 186 of its 1300 VALU operations per thread and tile).  The base
// goes through an opaque SGPR pair (as a global-address-space pointer: through a generic one the accesses would become flat_*).
// THE BASE MUST BE WAVE-UNIFORM (it may depend on blockIdx, the frame / chunk / tile of the loop, never on threadIdx): "+s" on a divergent value makes the
// compiler insert v_readfirstlane and every lane would silently use lane 0's address.  Build with -DINTFFT_AT32_CHECK (tools/evidence.sh <tag> at32check:
// every translation unit rebuilt, the parity suites run on that library) to have each at32 / at32b compare the base with its readfirstlane and trap.
template <typename T> using gptr_t = T __attribute__((address_space(1))) *;
__device__ __forceinline__ void at32_check(const void *uniform_base)
{
#ifdef INTFFT_AT32_CHECK
    const unsigned long long a = (unsigned long long)uniform_base;
    const unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)lo) != lo || (unsigned)__builtin_amdgcn_readfirstlane((int)hi) != hi) __builtin_trap();
#else
    (void)uniform_base;
#endif
}
template <typename T> __device__ __forceinline__ gptr_t<T> at32(T *uniform_base, unsigned elem_off)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    at32_check(uniform_base);
    gptr_t<B> g = (gptr_t<B>)uniform_base;
    asm("" : "+s"(g));
    return (gptr_t<T>)(g + (size_t)(elem_off * (unsigned)sizeof(T)));
}

// the same with the thread's BYTE offset given (one opaque 32-bit value shared by every access of a loop iteration)
template <typename T> __device__ __forceinline__ gptr_t<T> at32b(T *uniform_base, unsigned byte_off)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    at32_check(uniform_base);
    gptr_t<B> g = (gptr_t<B>)uniform_base;
    asm("" : "+s"(g));
    return (gptr_t<T>)(g + (size_t)byte_off);
}
__device__ __forceinline__ uint2 ld2_at32b(const uint2 *uniform_base, unsigned byte_off)
{
    typedef uint32_t v2u_t __attribute__((ext_vector_type(2)));
    const v2u_t x = *at32b(reinterpret_cast<const v2u_t *>(uniform_base), byte_off);
    return make_uint2(x.x, x.y);
}

// a uint2 table entry through at32 (HIP's vector structs do not copy out of an address-space-qualified reference)
__device__ __forceinline__ uint2 ld2_at32(const uint2 *uniform_base, unsigned elem_off)
{
    typedef uint32_t v2u_t __attribute__((ext_vector_type(2)));
    const v2u_t x = *at32(reinterpret_cast<const v2u_t *>(uniform_base), elem_off);
    return make_uint2(x.x, x.y);
}


// KIND_TWMUL / KIND_TWMULC: the inter-pass twiddle multiply of the N > 512K "2-D scheme" (DESIGN.md section 4.5):
// a pointwise Y = cmult(V, W_N^(k1*n2)) (forward) / T = V * conj(W) through the re/im-swapped feed (inverse)
enum StageKind : int { KIND_DIF = 0, KIND_DIT = 1, KIND_TWMUL = 2, KIND_TWMULC = 3 };
enum RoundKind : int { RND_TRUNC = 0, RND_ROUND = 1, RND_UNSCALED = 2 };

// One butterfly stage as the kernels see it (filled by the host planner).
struct StageDesc {
    int kind;    // KIND_DIF / KIND_DIT
    int s;       // index bit the butterfly pairs on (1-D plans: = the STAGE generic)
    int ts;      // STAGE generic of the core the stage belongs to: twiddle table, STAGE 0 / 1 special cases (1-D: = s)
    int tshift;  // twiddle index = (position mod 2^s) >> tshift (2-D scheme, column core: log2 N2; else 0)
    int lb;      // tile-local bit that carries index bit s
    int dtw;     // DTW: width of the stage inputs
    int wo;      // output width DTW - SCALE + 1
    int mw;      // width the complex multiplier works at (DIF: wo, DIT: dtw)
    int rnd;     // RoundKind
    int sh_a;    // multiplier: per-product pre-shift  (0 in the single-DSP regimes)
    int sh_b;    // multiplier: post-sum shift
    int narrow;  // (-3: set by a kernel at compile time, never by the planner: the three-dword product form of cmult<int64_t>)
                 // 1: 64-bit words with mw + TWDL_WIDTH <= 64, every product of the multiplier fits one int64;
                 // 59 / 61: trpl18 regime with mw beyond the multiplier's A port: the data operand is cut to that many bits first
                 // (SXT(M_AA, AWD), int_cmult_trpl18_dsp48.vhd:161-162); 0: neither
    unsigned tw_off; // offset of this stage's table in the twiddle buffer (int2 entries)
};

template <typename T> struct Cx { T re, im; };

// on-chip words: int32_t / int64_t, and __int128 for plans whose results exceed 64 bits (the trpl18 / trpl52 tails of
// int_cmult_dsp48.vhd:267-303, 396-433 with bit growth).  std::make_unsigned knows nothing of __int128 under -std=c++17.
typedef __int128 i128;
typedef unsigned __int128 u128;
template <typename T> struct UWord { using type = typename std::make_unsigned<T>::type; };
template <> struct UWord<i128> { using type = u128; };

// Ordering point for a WAVE-PRIVATE LDS tile: lanes of one wave exchange data through LDS (a ds_write, then a ds_read of
// another lane's word).  A wave's DS instructions issue and complete in order on gfx950; this states the ordering in the
// memory model instead of relying on that: release the writes at wavefront scope, keep the wave converged, acquire before
// the reads.  Lowers to no instructions beyond the lgkmcnt waits the compiler places anyway.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T> __device__ __forceinline__ T wrapw(T v, int w)
{
    constexpr int B = sizeof(T) * 8;
    if (w >= B) return v;
    using U = typename UWord<T>::type;
    return (T)((U)v << (B - w)) >> (B - w);
}

// "for positive values use Y = not(X) + 1, for negative values use Y = not(X)"
// int_dif2_fly.vhd:280-304, int_dit2_fly.vhd:251-276
template <typename T> __device__ __forceinline__ T neg_quirk(T x, int w)
{
    using U = typename UWord<T>::type;
    return x >= 0 ? wrapw<T>((T)((U)0 - (U)x), w) : (T)~x;
}

// ---- complex multiplier -------------------------------------------------------------------
// result = wrap_w( ((M2 >> a) -/+ (M1 >> a)) >> b ) with (a, b) per regime:
//   sngl (0, t-1) int_cmult_dsp48.vhd:189-190 | dbl18 (t-4|t-6, 3|5) int_cmult_dbl18_dsp48.vhd:163,174-175
//   trpl18 (t-1, 0) int_cmult_trpl18_dsp48.vhd:151-155 | sngl25 (0, t-2) int_cmult_dsp48.vhd:316-317
//   dbl35 (t-14, 12) int_cmult_dbl35_dsp48.vhd:163-168 | trpl52 (t-2, 0) int_cmult_trpl52_dsp48.vhd:166-170
// 32-bit words: products fit int64 (|d| < 2^31, |w| < 2^26).
__device__ __forceinline__ void cmult(int32_t dre, int32_t dim, int32_t wr, int32_t wi, int mw,
                                      int a, int b, int /*narrow*/, int32_t &ore, int32_t &oim)
{
    uint64_t m2r = (uint64_t)((int64_t)dre * wr), m1r = (uint64_t)((int64_t)dim * wi); // RE: M2 - M1  (:192-207)
    uint64_t m2i = (uint64_t)((int64_t)dre * wi), m1i = (uint64_t)((int64_t)dim * wr); // IM: M2 + M1  (:209-224)
    // (M >> a) << a == M & ~(2^a - 1), so ((M2 >> a) -/+ (M1 >> a)) >> b == ((M2 & K) -/+ (M1 & K)) >> (a + b) exactly; the
    // mw result bits start at bit a + b <= 26 of the sum: one v_alignbit_b32 and one v_bfe_i32 instead of three 64-bit shifts
    if (a) {
        const uint64_t k = 0xFFFFFFFF00000000ull | ~((1u << a) - 1u);
        m2r &= k, m1r &= k, m2i &= k, m1i &= k;
    }
    const uint64_t xr = m2r - m1r, xi = m2i + m1i;
    const uint32_t lr = __builtin_amdgcn_alignbit((uint32_t)(xr >> 32), (uint32_t)xr, (uint32_t)(a + b));
    const uint32_t li = __builtin_amdgcn_alignbit((uint32_t)(xi >> 32), (uint32_t)xi, (uint32_t)(a + b));
    ore = wrapw<int32_t>((int32_t)lr, mw);
    oim = wrapw<int32_t>((int32_t)li, mw);
}

// 64-bit words: products reach 2^89; keep each product as H*2^32 + L with
// H = (d >> 32) * w and L = (d & 0xffffffff) * w (both exact in int64), so that
// floor(M / 2^k) mod 2^64 = (H << (32 - k)) + (L >> k) for 0 <= k <= 32.  Only the low 64 bits of
// the sums are ever sliced by the RTL (w <= 64), so wrapping uint64 arithmetic is exact.
struct Prod96 { int64_t h, l; };
__device__ __forceinline__ Prod96 mul96(int64_t d, int32_t w)
{
    Prod96 p;
    p.h = (d >> 32) * (int64_t)w;
    p.l = (int64_t)(uint64_t)(uint32_t)d * (int64_t)w;
    return p;
}
__device__ __forceinline__ uint64_t shr96(Prod96 p, int k)
{
    return ((uint64_t)p.h << (32 - k)) + (uint64_t)(p.l >> k);
}
__device__ __forceinline__ void cmult(int64_t dre, int64_t dim, int32_t wr, int32_t wi, int mw,
                                      int a, int b, int narrow, int64_t &ore, int64_t &oim)
{
    if (narrow > 1) { // trpl18 beyond its A port: the operand is cut to `narrow` bits
        dre = wrapw<int64_t>(dre, narrow);
        dim = wrapw<int64_t>(dim, narrow);
    }
    if (narrow == 1) { // |d| <= 2^(mw-1), |w| <= 2^(t-1), mw + t <= 64: products and their sum are exact in int64
        // d = dH 2^32 + dL (dL signed): d * w = mad_i64_i32(dL, w) + (mul_lo(dH, w) << 32), exact modulo 2^64; the truncation points
        // as a mask ((M >> a) << a == M & ~(2^a - 1)); the mw result bits start at bit a + b <= 26 of the sum and are cut out
        // with v_alignbit_b32 / v_bfe_i32 where they lie inside one dword pair (else two 64-bit shifts)
        auto mul = [](int64_t d, int32_t w) -> uint64_t {
            const int32_t dl = (int32_t)d, dh = (int32_t)(d >> 32) - (dl >> 31);
            return (uint64_t)((int64_t)dl * w) + ((uint64_t)((uint32_t)dh * (uint32_t)w) << 32);
        };
        uint64_t m2r = mul(dre, wr), m1r = mul(dim, wi), m2i = mul(dre, wi), m1i = mul(dim, wr);
        if (a) {
            const uint64_t k = 0xFFFFFFFF00000000ull | ~((1u << a) - 1u);
            m2r &= k, m1r &= k, m2i &= k, m1i &= k;
        }
        const uint64_t xr = m2r - m1r, xi = m2i + m1i;
        const int sh = a + b;
        if (mw <= 32) {
            const uint32_t lr = __builtin_amdgcn_alignbit((uint32_t)(xr >> 32), (uint32_t)xr, (uint32_t)sh);
            const uint32_t li = __builtin_amdgcn_alignbit((uint32_t)(xi >> 32), (uint32_t)xi, (uint32_t)sh);
            ore = wrapw<int32_t>((int32_t)lr, mw);
            oim = wrapw<int32_t>((int32_t)li, mw);
        } else if (sh + mw <= 64 && sh + (mw - 32) <= 32) {
            const uint32_t lr = __builtin_amdgcn_alignbit((uint32_t)(xr >> 32), (uint32_t)xr, (uint32_t)sh);
            const uint32_t li = __builtin_amdgcn_alignbit((uint32_t)(xi >> 32), (uint32_t)xi, (uint32_t)sh);
            const int32_t hr = __builtin_amdgcn_sbfe((int32_t)(xr >> 32), sh, mw - 32);
            const int32_t hi = __builtin_amdgcn_sbfe((int32_t)(xi >> 32), sh, mw - 32);
            ore = (int64_t)(((uint64_t)(uint32_t)hr << 32) | lr);
            oim = (int64_t)(((uint64_t)(uint32_t)hi << 32) | li);
        } else {
            ore = wrapw<int64_t>((int64_t)xr >> sh, mw);
            oim = wrapw<int64_t>((int64_t)xi >> sh, mw);
        }
        return;
    }
    if (narrow == -3 || (mw <= 63 && a + b <= 31)) { // (-3: the caller has checked the condition for every stage of its plan)
        // three-dword products: d = dH 2^32 + dL (dL signed; |dH| < 2^31 for mw <= 63), P = (dH w + ((dL w) >> 32)) 2^32 + lo32(dL w): two
        // v_mad_i64_i32 per product; the truncation points as a mask on the low dword, the sums with carry chains, the mw result bits
        // from bit a + b <= 31 of the 96-bit sum with two v_alignbit_b32
        struct P3 { uint32_t w0, w1, w2; };
        auto mul = [](int64_t d, int32_t w) -> P3 {
            const int32_t dl = (int32_t)d, dh = (int32_t)(d >> 32) - (dl >> 31);
            const int64_t plo = (int64_t)dl * w;
            const int64_t u = (int64_t)dh * w + (plo >> 32);
            return P3{(uint32_t)plo, (uint32_t)u, (uint32_t)((uint64_t)u >> 32)};
        };
        P3 m2r = mul(dre, wr), m1r = mul(dim, wi), m2i = mul(dre, wi), m1i = mul(dim, wr);
        if (a) {
            const uint32_t k = ~((1u << a) - 1u);
            m2r.w0 &= k, m1r.w0 &= k, m2i.w0 &= k, m1i.w0 &= k;
        }
        auto sub = [](P3 x, P3 y) -> P3 {
            unsigned c0, c1, c2;
            P3 r;
            r.w0 = __builtin_subc(x.w0, y.w0, 0u, &c0);
            r.w1 = __builtin_subc(x.w1, y.w1, c0, &c1);
            r.w2 = __builtin_subc(x.w2, y.w2, c1, &c2);
            return r;
        };
        auto add = [](P3 x, P3 y) -> P3 {
            unsigned c0, c1, c2;
            P3 r;
            r.w0 = __builtin_addc(x.w0, y.w0, 0u, &c0);
            r.w1 = __builtin_addc(x.w1, y.w1, c0, &c1);
            r.w2 = __builtin_addc(x.w2, y.w2, c1, &c2);
            return r;
        };
        const P3 xr = sub(m2r, m1r), xi = add(m2i, m1i);
        const uint32_t sh = (uint32_t)(a + b);
        const uint32_t lr = __builtin_amdgcn_alignbit(xr.w1, xr.w0, sh), hr = __builtin_amdgcn_alignbit(xr.w2, xr.w1, sh);
        const uint32_t li = __builtin_amdgcn_alignbit(xi.w1, xi.w0, sh), hi = __builtin_amdgcn_alignbit(xi.w2, xi.w1, sh);
        ore = wrapw<int64_t>((int64_t)(((uint64_t)hr << 32) | lr), mw);
        oim = wrapw<int64_t>((int64_t)(((uint64_t)hi << 32) | li), mw);
        return;
    }
    const Prod96 m2r = mul96(dre, wr), m1r = mul96(dim, wi);
    const Prod96 m2i = mul96(dre, wi), m1i = mul96(dim, wr);
    uint64_t r, i;
    if (a == 0) { // exact sum first, then one slice
        Prod96 sr{m2r.h - m1r.h, m2r.l - m1r.l}, si{m2i.h + m1i.h, m2i.l + m1i.l};
        r = shr96(sr, b);
        i = shr96(si, b);
    } else {
        r = (uint64_t)((int64_t)(shr96(m2r, a) - shr96(m1r, a)) >> b);
        i = (uint64_t)((int64_t)(shr96(m2i, a) + shr96(m1i, a)) >> b);
    }
    ore = wrapw<int64_t>((int64_t)r, mw);
    oim = wrapw<int64_t>((int64_t)i, mw);
}

// 128-bit words: |d| < 2^(mw-1) with mw <= 96, |w| < 2^27: every product and sum is exact in 128 bits
__device__ __forceinline__ void cmult(i128 dre, i128 dim, int32_t wr, int32_t wi, int mw, int a, int b, int narrow,
                                      i128 &ore, i128 &oim)
{
    if (narrow > 1) { // trpl18 beyond its A port (see StageDesc::narrow)
        dre = wrapw<i128>(dre, narrow);
        dim = wrapw<i128>(dim, narrow);
    }
    const i128 m2r = dre * (i128)wr, m1r = dim * (i128)wi;
    const i128 m2i = dre * (i128)wi, m1i = dim * (i128)wr;
    ore = wrapw<i128>(((m2r >> a) - (m1r >> a)) >> b, mw);
    oim = wrapw<i128>(((m2i >> a) + (m1i >> a)) >> b, mw);
}

// ---- sum / difference with the three scaling variants --------------------------------------
// trunc  : (A >> 1) +/- (B >> 1)   LSB dropped BEFORE the add (int_dif2_fly.vhd:151-154)
// round  : rhu2(A +/- B) on the exact (DTW+1)-bit sum, wrapped to DTW bits (:173-218); written
//          without the extra bit: rhu2(A+B) = (A|B) - ((A^B) >> 1), rhu2(A-B) = rhu2(A+B) - B  (derivation: sumdiff, intfft_pk16.hpp)
// unscaled: A +/- B, one bit of growth (:222-240)
// RNDC: the rounding kind when the caller knows it at compile time (k_pass<T, RND>: one kind per plan), else -1
template <typename T, int RNDC = -1>
__device__ __forceinline__ void addsub(T a, T b, int rnd_rt, int wo, T &s, T &d)
{
    using U = typename UWord<T>::type;
    const int rnd = RNDC >= 0 ? RNDC : rnd_rt;
    if (rnd == RND_TRUNC) {
        s = (a >> 1) + (b >> 1);
        d = (a >> 1) - (b >> 1);
    } else if (rnd == RND_ROUND) {
        s = wrapw<T>((T)((U)(a | b) - (U)((a ^ b) >> 1)), wo); // (the wrap of s is the identity on in-range operands; kept here: generic kernel)
        d = wrapw<T>((T)((U)s - (U)b), wo);                       // rhu2(A - B) = rhu2(A + B) - B
    } else {
        s = (T)((U)a + (U)b);
        d = (T)((U)a - (U)b);
    }
}

// int_dif2_fly: X = S, Y = D (STAGE 0) | D * {1, -j} (STAGE 1) | cmult(D, W) (STAGE >= 2)
// CLS: what the caller knows about the stage at compile time: 0 nothing, 1 STAGE < 2 (no multiplier), 2 STAGE >= 2
template <typename T, int RNDC = -1, int CLS = 0>
__device__ __forceinline__ void dif_fly(const StageDesc &st, int odd, Cx<T> a, Cx<T> b, int32_t wr,
                                        int32_t wi, Cx<T> &x, Cx<T> &y)
{
    Cx<T> s, d;
    addsub<T, RNDC>(a.re, b.re, st.rnd, st.wo, s.re, d.re);
    addsub<T, RNDC>(a.im, b.im, st.rnd, st.wo, s.im, d.im);
    x = s;
    if (CLS != 2 && st.ts == 0) { // int_dif2_fly.vhd:245-255
        y = d;
    } else if (CLS != 2 && (CLS == 1 || st.ts == 1)) { // :259-318
        if (!odd) {
            y = d;
        } else {
            y.re = d.im;
            y.im = neg_quirk<T>(d.re, st.wo);
        }
    } else { // :322-373
        cmult(d.re, d.im, wr, wi, st.mw, st.sh_a, st.sh_b, st.narrow, y.re, y.im);
    }
}

// int_dit2_fly: T = B (STAGE 0) | B * {1, +j} (STAGE 1) | B * conj(W) via the re/im-swapped
// multiplier (int_dit2_fly.vhd:304-322), then X = A + T, Y = A - T with the scaling variant.
template <typename T, int RNDC = -1, int CLS = 0>
__device__ __forceinline__ void dit_fly(const StageDesc &st, int odd, Cx<T> a, Cx<T> b, int32_t wr,
                                        int32_t wi, Cx<T> &x, Cx<T> &y)
{
    Cx<T> t;
    if (CLS != 2 && st.ts == 0) { // :221-230
        t = b;
    } else if (CLS != 2 && (CLS == 1 || st.ts == 1)) { // :234-286
        if (!odd) {
            t = b;
        } else {
            t.im = b.re;
            t.re = neg_quirk<T>(b.im, st.dtw);
        }
    } else {
        T ore, oim;
        cmult(b.im, b.re, wr, wi, st.mw, st.sh_a, st.sh_b, st.narrow, ore, oim);
        t.im = ore;
        t.re = oim;
    }
    addsub<T, RNDC>(a.re, t.re, st.rnd, st.wo, x.re, y.re);
    addsub<T, RNDC>(a.im, t.im, st.rnd, st.wo, x.im, y.im);
}

// ---- 2-D scheme: the inter-pass twiddle W_N^m, evaluated on the fly -------------------------------------
// The quarter-wave ROM formula of rom_twiddle_int.vhd:143-152 at full depth, (re, im) = (RN(mg cos phi), RN(mg sin(-phi))),
// phi = 2 pi a / N for a = m mod N/4, rotated by the quadrant rule (re, im) <- (im, -re) of :177-183 (m div N/4 times).  So that
// every implementation (this one on the GPU and on the host, the C oracle, the Python twin) produces the SAME integers, cos and sin
// are a fixed sequence of IEEE double operations, each rounded separately (no fused multiply-add): octant reduction to
// x in [0, pi/4], x = b * (pi * 2^-(L-1)), Taylor polynomials in z = x*x up to z^8 in Horner form (truncation < 3e-18;
// |error| < 2e-16), RN(v) = floor(v + 0.5) on the non-negative first-octant values.  A table of N entries would cost one random
// 128-byte line per sample (measured: 3x the time of the whole rest of the plan); this costs ~40 double operations.
// the two constants of a plan: pi * 2^-(L-1) (M_PI scaled by a power of two: exact) and mg = 2^(t-1) - 1 (t < 18) or 2^(t-2) - 1
// (rom_twiddle_int.vhd:143-147; exact)
__host__ __device__ inline void tw2d_consts(int L, int t, double &scale, double &mg)
{
#pragma clang fp contract(off)
    scale = 3.14159265358979323846;
    for (int i = 0; i < L - 1; ++i) scale = scale * 0.5;
    mg = 1.0;
    for (int i = 0; i < (t < 18 ? t - 1 : t - 2); ++i) mg = mg * 2.0;
    mg = mg - 1.0;
}

__host__ __device__ inline void tw2d_eval(int L, double scale, double mg, unsigned m, int &re, int &im)
{
#pragma clang fp contract(off)
    const unsigned quarter = 1u << (L - 2);
    const unsigned a = m & (quarter - 1u), q = (m >> (L - 2)) & 3u;
    const bool swap = a > (quarter >> 1);
    const unsigned b = swap ? quarter - a : a;
    const double x = (double)b * scale;
    const double z = x * x;
    double pc = 0x1.ae7f3e733b81fp-45;
    pc = -0x1.93974a8c07c9dp-37 + z * pc;
    pc = 0x1.1eed8eff8d898p-29 + z * pc;
    pc = -0x1.27e4fb7789f5cp-22 + z * pc;
    pc = 0x1.a01a01a01a01ap-16 + z * pc;
    pc = -0x1.6c16c16c16c17p-10 + z * pc;
    pc = 0x1.5555555555555p-5 + z * pc;
    pc = -0x1.0000000000000p-1 + z * pc;
    const double cosx = 1.0 + z * pc;
    double ps = 0x1.952c77030ad4ap-49;
    ps = -0x1.ae7f3e733b81fp-41 + z * ps;
    ps = 0x1.6124613a86d09p-33 + z * ps;
    ps = -0x1.ae64567f544e4p-26 + z * ps;
    ps = 0x1.71de3a556c734p-19 + z * ps;
    ps = -0x1.a01a01a01a01ap-13 + z * ps;
    ps = 0x1.1111111111111p-7 + z * ps;
    ps = -0x1.5555555555555p-3 + z * ps;
    const double xz = x * z;
    const double sinx = x + xz * ps;
    const double vc = mg * (swap ? sinx : cosx) + 0.5, vs = mg * (swap ? cosx : sinx) + 0.5;
    int c = (int)(long long)vc, sn = -(int)(long long)vs; // first quadrant: (RN(mg cos), -RN(mg sin))
    // (re, im) <- (im, -re), q times (|values| <= mg < 2^(t-1): the negation fits t bits)
    re = q == 0 ? c : q == 1 ? sn : q == 2 ? -c : -sn;
    im = q == 0 ? sn : q == 1 ? -c : q == 2 ? -sn : c;
}

__host__ __device__ inline void tw2d_eval(int L, int t, unsigned m, int &re, int &im)
{
    double scale, mg;
    tw2d_consts(L, t, scale, mg);
    tw2d_eval(L, scale, mg, m, re, im);
}

// ---- I/O order maps (include/intfft.h) -------------------------------------------------------
__device__ __forceinline__ unsigned brev_l(unsigned v, int L) { return __brev(v) >> (32 - L); }

// memory index of logical index `idx` for one of the INTFFT_ORDER_* layouts (inverse of the
// "memory -> logical" maps documented in intfft.h)
__device__ __forceinline__ unsigned order_to_mem(int order, int L, unsigned idx)
{
    const unsigned half = 1u << (L - 1);
    switch (order) {
    case 1: return brev_l(idx, L);                                   // BITREV
    case 2: return ((idx & (half - 1)) << 1) | (idx >> (L - 1));     // HALVES: rotate left
    case 3: {                                                         // BITREV_LANES
        const unsigned r = brev_l(idx, L);                            // = 2*(m mod N/2) + m div N/2
        return (r >> 1) | ((r & 1u) << (L - 1));
    }
    default: return idx;                                             // NATURAL
    }
}

// logical index stored at memory index m (the maps documented in intfft.h)
__device__ __forceinline__ unsigned order_from_mem(int order, int L, unsigned m)
{
    const unsigned half = 1u << (L - 1);
    switch (order) {
    case 1: return brev_l(m, L);                                              // BITREV
    case 2: return (m >> 1) + (m & 1u) * half;                                // HALVES
    case 3: return brev_l(2u * (m & (half - 1)) + (m >> (L - 1)), L);         // BITREV_LANES
    default: return m;                                                        // NATURAL
    }
}

} // namespace intfft
