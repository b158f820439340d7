// intfft_generic.hip -- generic LDS pass kernel + twiddle-generation kernels (gfx950).
//
// k_pass<T> evaluates any run of radix-2 stages of int_fftNk / int_ifftNk
// (src/vhdl/fft/int_fftNk.vhd:184-342, src/vhdl/fft/int_ifftNk.vhd:183-341) on a tile held in LDS,
// for every width/mode/regime the RTL elaborates.  The cross-commutators
// (src/vhdl/delay/int_delay_line.vhd:60-104) never materialise: they are what makes the flat
// in-place index of SURVEY.md section 9.1 valid.  The packed-int16 wave kernel in intfft_fast1024.hip is the
// speed path for the headline configuration; this kernel is the complete one.
#include "intfft_internal.hpp"

#include <algorithm>
#include <cstdlib>

#include <cmath>

namespace intfft {

// LDS index padding: one element of padding per 32 keeps the bit-reversed / strided tile sweeps
// of the load and store phases from landing on one bank.
__device__ __forceinline__ unsigned pad(unsigned e) { return e + (e >> 5); }

template <typename T> __device__ __forceinline__ T load_user(const void *base, int cb, size_t i)
{
    if (cb == 2) return (T) reinterpret_cast<const int16_t *>(base)[i];
    if (cb == 4) return (T) reinterpret_cast<const int32_t *>(base)[i];
    if (cb == 16) return (T) reinterpret_cast<const i128 *>(base)[i];
    return (T) reinterpret_cast<const int64_t *>(base)[i];
}

template <typename T> __device__ __forceinline__ void store_user(void *base, int cb, size_t i, T v)
{
    if (cb == 2) reinterpret_cast<int16_t *>(base)[i] = (int16_t)v;
    else if (cb == 4) reinterpret_cast<int32_t *>(base)[i] = (int32_t)v;
    else if (cb == 16) reinterpret_cast<i128 *>(base)[i] = (i128)v; // little-endian two's complement, low word first
    else reinterpret_cast<int64_t *>(base)[i] = (int64_t)v;
}

// 2^R-point sub-transform in registers: element r of the group lives at lds[pad(e + (r << lb_lo))] and
// carries index bits s_lo .. s_lo+R-1 = r.  Stage s_lo+i pairs r-bit i and uses twiddle
// kb + ((r mod 2^i) << s_lo) of its table; stage descriptors a.st[si ..] are in processing order.
// NARROW (64-bit words): every multiplier stage of the pass has StageDesc::narrow == 1, so the regime test of cmult() folds at
// compile time and the 96-bit product path leaves the instruction stream (as in intfft_fastw64.hip)
template <typename T, int KIND, int R, int RND, bool NARROW>
__device__ __forceinline__ void round_generic(Cx<T> *lds, unsigned e, int lb_lo, int s_lo, unsigned kb,
                                              const PassArgs &a, int si, const int2 *__restrict__ tw)
{
    constexpr int P = 1 << R;
    Cx<T> v[P];
#pragma unroll
    for (int r = 0; r < P; ++r) v[r] = lds[pad(e + ((unsigned)r << lb_lo))];
#pragma unroll
    for (int ii = 0; ii < R; ++ii) {
        const int i = KIND == KIND_DIF ? R - 1 - ii : ii;
        StageDesc st = a.st[si + ii];
        if (NARROW) st.narrow = 1;
        const int h = 1 << i;
        // the kind of the stage (multiplier / STAGE 1 / STAGE 0) is uniform: one branch per stage, straight-line butterflies inside
        if (st.ts >= 2) {
#pragma unroll
            for (int pr = 0; pr < P / 2; ++pr) {
                const int r0 = ((pr >> i) << (i + 1)) | (pr & (h - 1));
                const int2 w = tw[st.tw_off + ((kb + ((unsigned)(r0 & (h - 1)) << s_lo)) >> st.tshift)];
                Cx<T> X, Y;
                if (KIND == KIND_DIF) dif_fly<T, RND, 2>(st, 0, v[r0], v[r0 + h], w.x, w.y, X, Y);
                else dit_fly<T, RND, 2>(st, 0, v[r0], v[r0 + h], w.x, w.y, X, Y);
                v[r0] = X;
                v[r0 + h] = Y;
            }
        } else {
#pragma unroll
            for (int pr = 0; pr < P / 2; ++pr) {
                const int r0 = ((pr >> i) << (i + 1)) | (pr & (h - 1));
                const unsigned k = (kb + ((unsigned)(r0 & (h - 1)) << s_lo)) >> st.tshift;
                Cx<T> X, Y;
                if (KIND == KIND_DIF) dif_fly<T, RND, 1>(st, (int)(k & 1u), v[r0], v[r0 + h], 0, 0, X, Y);
                else dit_fly<T, RND, 1>(st, (int)(k & 1u), v[r0], v[r0 + h], 0, 0, X, Y);
                v[r0] = X;
                v[r0 + h] = Y;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < P; ++r) lds[pad(e + ((unsigned)r << lb_lo))] = v[r];
}

// RND: the rounding kind of the plan's butterflies (one per plan: FORMAT / RNDMODE) as a template parameter -- a third of the
// sum / difference code per instantiation, no run-time mode branches in the register rounds
template <typename T, int RND, bool NARROW = false>
__global__ __launch_bounds__(1024) void k_pass(const PassArgs a, const void *in, void *out,
                                                       const int2 *__restrict__ tw, size_t nframes,
                                                       const int2 *__restrict__ tw2d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cx<T> *lds = reinterpret_cast<Cx<T> *>(smem);

    const int L = a.L, U = a.U;
    const unsigned tiles = 1u << (L - U);
    const unsigned tile = blockIdx.x % tiles;
    const size_t f0 = (size_t)(blockIdx.x / tiles) * (size_t)a.fpb;
    const unsigned nf = (unsigned)min((size_t)a.fpb, nframes - f0);
    const size_t N = (size_t)1 << L;

    // deposit the tile id into the index bits the tile does not own
    unsigned tile_bits = 0;
    {
        unsigned t = tile;
        for (int b = 0; b < L; ++b) {
            const bool owned = (b >= a.pos0 && b < a.pos0 + a.len0) || (b >= a.pos1 && b < a.pos1 + a.len1);
            if (!owned) {
                tile_bits |= (t & 1u) << b;
                t >>= 1;
            }
        }
    }
    const unsigned m0 = (1u << a.len0) - 1u;
    auto spread = [&](unsigned u) -> unsigned {
        return tile_bits | ((u & m0) << a.pos0) | ((u >> a.len0) << a.pos1);
    };
    auto swap_runs = [&](unsigned v) -> unsigned { // v enumerates run1 fastest
        const unsigned m1 = (1u << a.len1) - 1u;
        return ((v & m1) << a.len0) | (v >> a.len1);
    };

    const unsigned tile_n = 1u << U;
    const unsigned total = nf << U;

    // ---- load -------------------------------------------------------------------------------
    for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
        const unsigned f = i >> U, v = i & (tile_n - 1u);
        unsigned u = a.ld_swap ? swap_runs(v) : v;
        unsigned j = spread(u);
        Cx<T> c;
        if (a.in_mode == IO_USER) {
            size_t m;
            if (a.ld_memorder) { // v is the memory index; the tile owns every bit, so u == j
                const unsigned logical = order_from_mem(a.in_order, L, v);
                u = j = a.in_rev ? brev_l(logical, L) : logical;
                m = (f0 + f) * N + v;
            } else {
                const unsigned logical = a.in_rev ? brev_l(j, L) : j;
                m = (f0 + f) * N + order_to_mem(a.in_order, L, logical);
            }
            c.re = load_user<T>(in, a.in_cb, 2 * m);
            c.im = load_user<T>(in, a.in_cb, 2 * m + 1);
            if (a.in_zext) {
                using UT = typename UWord<T>::type;
                const UT mask = (a.in_bits >= (int)(8 * sizeof(T))) ? ~(UT)0 : (((UT)1 << a.in_bits) - 1);
                c.re = (T)((UT)c.re & mask);
                c.im = (T)((UT)c.im & mask);
            } else {
                c.re = wrapw<T>(c.re, a.in_bits);
                c.im = wrapw<T>(c.im, a.in_bits);
            }
        } else if (sizeof(T) == 8 && a.scr_in_word == 4) { // written by a narrower (int32) pass
            const Cx<int32_t> n = reinterpret_cast<const Cx<int32_t> *>(in)[(f0 + f) * N + j];
            c.re = n.re;
            c.im = n.im;
        } else {
            c = reinterpret_cast<const Cx<T> *>(in)[(f0 + f) * N + j];
        }
        lds[pad((f << U) + u)] = c;
    }
    __syncthreads();

    // ---- stages: up to RMAX consecutive stages per LDS round trip, evaluated in registers ----------
    // Stages of one pass act on consecutive tile-local bits (descending for DIF, ascending for DIT), so a run
    // of R stages is a 2^R-point sub-transform per thread (round_generic<>), each stage with its own widths.
    // (int32 round mode: four stages of 16 points with the rhu2 temporaries spill under the 128-register bound of 1024 threads)
    constexpr int RMAX = sizeof(T) == 4 ? (RND == RND_ROUND ? 3 : 4) : sizeof(T) == 8 ? 3 : 2;
    int si = 0;
    while (si < a.nstages) {
        const StageDesc st = a.st[si];
        if (st.kind == KIND_TWMUL || st.kind == KIND_TWMULC) {
            // 2-D scheme, between the cores: element j = (j1, n2) is multiplied by W_N^(rev(j1) * n2), tshift = log2 N2
            const int l2 = st.tshift, l1 = L - l2;
            for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
                const unsigned f = i >> U, u = i & (tile_n - 1u);
                const unsigned j = spread(u);
                const unsigned m = (brev_l(j >> l2, l1) * (j & ((1u << l2) - 1u))) & ((1u << L) - 1u);
                const int2 w = tw2d[m];
                Cx<T> &c = lds[pad((f << U) + u)];
                T ore, oim;
                if (st.kind == KIND_TWMUL) {
                    cmult(c.re, c.im, w.x, w.y, st.mw, st.sh_a, st.sh_b, st.narrow, ore, oim);
                    c.re = ore, c.im = oim;
                } else { // swapped feed (int_dit2_fly.vhd:304-322)
                    cmult(c.im, c.re, w.x, w.y, st.mw, st.sh_a, st.sh_b, st.narrow, ore, oim);
                    c.im = ore, c.re = oim;
                }
            }
            __syncthreads();
            ++si;
            continue;
        }
        int R = 1;
        while (R < RMAX && si + R < a.nstages && a.st[si + R].kind == st.kind && a.st[si + R].tshift == st.tshift &&
               a.st[si + R].lb == st.lb + (st.kind == KIND_DIF ? -R : R))
            ++R;
        const int lb_lo = st.kind == KIND_DIF ? st.lb - (R - 1) : st.lb;
        const int s_lo = st.kind == KIND_DIF ? st.s - (R - 1) : st.s;
        const unsigned ngroups = nf << (U - R);
        const unsigned lowm = (1u << lb_lo) - 1u;
        for (unsigned g = threadIdx.x; g < ngroups; g += blockDim.x) {
            const unsigned f = g >> (U - R), gg = g & ((tile_n >> R) - 1u);
            const unsigned u0 = ((gg >> lb_lo) << (lb_lo + R)) | (gg & lowm);
            const unsigned kb = spread(u0) & ((1u << s_lo) - 1u);
            const unsigned e = (f << U) + u0;
            if (st.kind == KIND_DIF) {
                switch (R) {
                case 4: if constexpr (RMAX >= 4) round_generic<T, KIND_DIF, 4, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                case 3: round_generic<T, KIND_DIF, 3, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                case 2: round_generic<T, KIND_DIF, 2, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                default: round_generic<T, KIND_DIF, 1, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                }
            } else {
                switch (R) {
                case 4: if constexpr (RMAX >= 4) round_generic<T, KIND_DIT, 4, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                case 3: round_generic<T, KIND_DIT, 3, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                case 2: round_generic<T, KIND_DIT, 2, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                default: round_generic<T, KIND_DIT, 1, RND, NARROW>(lds, e, lb_lo, s_lo, kb, a, si, tw); break;
                }
            }
        }
        __syncthreads();
        si += R;
    }

    // ---- store ------------------------------------------------------------------------------
    for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
        const unsigned f = i >> U, v = i & (tile_n - 1u);
        unsigned u = a.st_swap ? swap_runs(v) : v;
        unsigned j = spread(u);
        size_t m = 0;
        if (a.out_mode == IO_USER) {
            if (a.st_memorder) {
                const unsigned logical = order_from_mem(a.out_order, L, v);
                u = j = a.out_rev ? brev_l(logical, L) : logical;
                m = (f0 + f) * N + v;
            } else {
                const unsigned logical = a.out_rev ? brev_l(j, L) : j;
                m = (f0 + f) * N + order_to_mem(a.out_order, L, logical);
            }
        }
        const Cx<T> c = lds[pad((f << U) + u)];
        if (a.out_mode == IO_USER) {
            store_user<T>(out, a.out_cb, 2 * m, c.re);
            store_user<T>(out, a.out_cb, 2 * m + 1, c.im);
        } else {
            reinterpret_cast<Cx<T> *>(out)[(f0 + f) * N + j] = c;
        }
    }
}

// big tiles leave room for only 2 workgroups per CU: give those workgroups 1024 threads
unsigned pass_threads(const PassArgs &a)
{
    const size_t elems = (size_t)a.fpb << a.U;
    static const int mode = diag_env("INTFFT_PASS_THREADS") ? atoi(diag_env("INTFFT_PASS_THREADS")) : 1;
    if (mode == 1) { // one thread per register round group: 16 points (int32 words) / 8 points (int64 words)
        const size_t t = elems / (a.word == 4 ? 16 : a.word == 8 ? 8 : 4);
        return (unsigned)(t < 256 ? 256 : t > 1024 ? 1024 : t);
    }
    return elems >= 8192 ? 1024u : elems >= 4096 ? 512u : (unsigned)PASS_THREADS;
}

size_t pass_lds_bytes(const PassArgs &a, int word_bytes)
{
    const size_t elems = (size_t)a.fpb << a.U;
    return (elems + (elems >> 5) + 1) * 2 * (size_t)word_bytes;
}

const char *pass_kernel_name(int word_bytes)
{
    return word_bytes == 4 ? "k_pass<int>" : word_bytes == 8 ? "k_pass<long>" : "k_pass<__int128>";
}

hipError_t launch_pass(const PassArgs &a, int word_bytes, const void *in, void *out, const int2 *tw,
                       size_t nframes, hipStream_t stream, const int2 *tw2d)
{
    if (nframes == 0) return hipSuccess;
    const size_t groups = (nframes + (size_t)a.fpb - 1) / (size_t)a.fpb;
    const size_t blocks = groups << (a.L - a.U);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = pass_lds_bytes(a, word_bytes);
    int rnd = RND_UNSCALED; // the butterflies of a plan share one kind (the 2-D scheme's multiplier stages carry none)
    bool narrow64 = word_bytes == 8 && !diag_env("INTFFT_NO_NARROW_PASS");
    for (int i = 0; i < a.nstages; ++i)
        if (a.st[i].kind == KIND_DIF || a.st[i].kind == KIND_DIT) {
            rnd = a.st[i].rnd;
            if (a.st[i].ts >= 2 && a.st[i].narrow != 1) narrow64 = false;
        }
#define INTFFT_LAUNCH_PASS(T, R, ...)                                                                                     \
    {                                                                                                                     \
        allow_max_lds(kptr(&k_pass<T, R, ##__VA_ARGS__>));                                                                \
        hipLaunchKernelGGL((k_pass<T, R, ##__VA_ARGS__>), dim3((unsigned)blocks), dim3(pass_threads(a)), lds, stream, a, in, out, tw, nframes, tw2d); \
    }
#define INTFFT_LAUNCH_PASS_T(T)                                                                                           \
    {                                                                                                                     \
        if (rnd == RND_TRUNC) INTFFT_LAUNCH_PASS(T, RND_TRUNC)                                                            \
        else if (rnd == RND_ROUND) INTFFT_LAUNCH_PASS(T, RND_ROUND)                                                       \
        else INTFFT_LAUNCH_PASS(T, RND_UNSCALED)                                                                          \
    }
    if (word_bytes == 4) INTFFT_LAUNCH_PASS_T(int32_t)
    else if (word_bytes == 16) INTFFT_LAUNCH_PASS(i128, RND_UNSCALED) // results beyond 64 bits exist with bit growth only
    else if (narrow64) {
        if (rnd == RND_TRUNC) INTFFT_LAUNCH_PASS(int64_t, RND_TRUNC, true)
        else if (rnd == RND_ROUND) INTFFT_LAUNCH_PASS(int64_t, RND_ROUND, true)
        else INTFFT_LAUNCH_PASS(int64_t, RND_UNSCALED, true)
    } else INTFFT_LAUNCH_PASS_T(int64_t)
#undef INTFFT_LAUNCH_PASS_T
#undef INTFFT_LAUNCH_PASS
    return hipGetLastError();
}

// ---- 2-D scheme: the multiplier between the cores ------------------------------------------------------
// In place on the [k1][n2] layout (element idx: k1 = idx >> l2, n2 = idx mod 2^l2): V <- int_cmult_dsp48(V, W_N^(k1 n2)) at width
// mw (forward), or T = V * conj(W) through the re/im-swapped multiplier feed of int_dit2_fly.vhd:304-322 (inverse).
template <typename T, typename PK> struct PairIO; // one (re, im) sample as a single load / store
template <> struct PairIO<int32_t, uint32_t> { // int16 containers
    static __device__ __forceinline__ void get(uint32_t v, int32_t &re, int32_t &im) { re = (int16_t)(v & 0xFFFFu), im = (int16_t)(v >> 16); }
    static __device__ __forceinline__ uint32_t put(int32_t re, int32_t im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }
};
template <> struct PairIO<int32_t, uint2> {
    static __device__ __forceinline__ void get(uint2 v, int32_t &re, int32_t &im) { re = (int32_t)v.x, im = (int32_t)v.y; }
    static __device__ __forceinline__ uint2 put(int32_t re, int32_t im) { return make_uint2((uint32_t)re, (uint32_t)im); }
};
template <> struct PairIO<int64_t, ulonglong2> {
    static __device__ __forceinline__ void get(ulonglong2 v, int64_t &re, int64_t &im) { re = (int64_t)v.x, im = (int64_t)v.y; }
    static __device__ __forceinline__ ulonglong2 put(int64_t re, int64_t im) { return make_ulonglong2((unsigned long long)re, (unsigned long long)im); }
};

// One thread owns one position (k1, n2) and walks the frames of the launch: its twiddle is evaluated ONCE (about 40 double
// operations) and applied to every frame -- per sample the kernel is a 16-bit load, one multiplier and a store.
template <typename T, typename PK>
__global__ __launch_bounds__(256) void k_twmul(PK *data, int L, int l2, int mw, int sh_a, int sh_b, int narrow, int conj, int twd,
                                               size_t nframes, unsigned fsplit)
{
    const unsigned nmask = (1u << L) - 1u, m2 = (1u << l2) - 1u;
    const size_t n = (size_t)1 << L;
    const unsigned idx = (blockIdx.x / fsplit) * 256u + threadIdx.x; // position; blocks with the same position share the frames
    const unsigned part = blockIdx.x % fsplit;
    if (idx > nmask) return;
    double scale, mg;
    tw2d_consts(L, twd, scale, mg);
    int wr, wi;
    tw2d_eval(L, scale, mg, ((idx >> l2) * (idx & m2)) & nmask, wr, wi);
    constexpr int UNR = 8; // independent frames per step: the loads are issued together
    for (size_t f0 = part; f0 < nframes; f0 += (size_t)fsplit * UNR) {
        PK d[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t f = f0 + (size_t)u * fsplit;
            if (f < nframes) d[u] = data[f * n + idx];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t f = f0 + (size_t)u * fsplit;
            if (f < nframes) {
                T re, im, ore, oim;
                PairIO<T, PK>::get(d[u], re, im);
                if (!conj) {
                    cmult(re, im, wr, wi, mw, sh_a, sh_b, narrow, ore, oim);
                    data[f * n + idx] = PairIO<T, PK>::put(ore, oim);
                } else {
                    cmult(im, re, wr, wi, mw, sh_a, sh_b, narrow, ore, oim);
                    data[f * n + idx] = PairIO<T, PK>::put(oim, ore);
                }
            }
        }
    }
}

hipError_t launch_twmul(void *data, int container_bytes, int L, int l2, int mw, int sh_a, int sh_b, int narrow, int conj,
                        int twd, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    const size_t pos_blocks = (((size_t)1 << L) + 255) / 256;
    // enough blocks to fill the chip: split the frames of a position over `fsplit` blocks when there are few positions
    unsigned fsplit = 1;
    const size_t want = (size_t)device_cus() * 16;
    while (pos_blocks * fsplit < want && fsplit * 2 <= nframes) fsplit *= 2;
    const unsigned blocks = (unsigned)(pos_blocks * fsplit);
    if (container_bytes == 2)
        hipLaunchKernelGGL((k_twmul<int32_t, uint32_t>), dim3(blocks), dim3(256), 0, stream, static_cast<uint32_t *>(data), L, l2, mw, sh_a,
                           sh_b, narrow, conj, twd, nframes, fsplit);
    else if (container_bytes == 4)
        hipLaunchKernelGGL((k_twmul<int32_t, uint2>), dim3(blocks), dim3(256), 0, stream, static_cast<uint2 *>(data), L, l2, mw, sh_a,
                           sh_b, narrow, conj, twd, nframes, fsplit);
    else
        hipLaunchKernelGGL((k_twmul<int64_t, ulonglong2>), dim3(blocks), dim3(256), 0, stream, static_cast<ulonglong2 *>(data), L, l2, mw,
                           sh_a, sh_b, narrow, conj, twd, nframes, fsplit);
    return hipGetLastError();
}

// ---- twiddle generation ---------------------------------------------------------------------
// k_twiddle_stage: the stream rom_twiddle_int emits for cnt = 0 .. 2^stage-1.
//   d_rom: the 512-entry quarter-wave ROM (DEPTH = 9) of integer (cos, -sin) at width twd; the
//   smaller ROMs of STAGE < 11 are exact sub-samplings of it (phi = ii*pi/2^(DEPTH+1),
//   rom_twiddle_int.vhd:149).  The double-precision cos/sin seeds are computed on the host
//   (intfft_plan.hip) like the RTL's elaboration-time constants; everything integer happens here.
__global__ void k_twiddle_stage(const int2 *__restrict__ rom, int stage, int twd, int xs,
                                long long mathpi, int2 *__restrict__ out)
{
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << stage)) return;
    if (stage == 0) {
        out[0] = rom[0];
        return;
    }
    const unsigned div = k >> (stage - 1);                  // cnt(STAGE-1)      rom_twiddle_int.vhd:189
    const unsigned addr = k & ((1u << (stage - 1)) - 1u);   // cnt(STAGE-2..0)   :188
    int2 e;
    unsigned cnt = 0;
    if (stage < 11) {
        e = rom[addr << (10 - stage)];                      // xSTD :205-212
    } else {
        e = rom[addr >> (stage - 10)];                      // addrx :221
        cnt = addr & ((1u << (stage - 10)) - 1u);           // count :225
    }
    if (div) { // second quadrant (re, im) <- (im, -re)  :177-183
        const int re = e.y, im = wrapw<int32_t>(-e.x, twd);
        e.x = re;
        e.y = im;
    }
    if (stage >= 11) { // row_twiddle_tay.vhd:123-268
        // xs = XSHIFT (find_widthA :123-133), mathpi = MATHPI (const_pi :135-149): host constants
        const long long mpi = (mathpi * (long long)cnt) & 0xFFFF; // :208-221
        const long long mpx = mpi >> 1;                            // :247
        long long p_re = (long long)e.x * (1ll << xs) + (long long)e.y * mpx; // MULT_SUB: C + A*B
        long long p_im = (long long)e.y * (1ll << xs) - (long long)e.x * mpx; // MULT_ADD: C - A*B
        p_re = wrapw<int64_t>(p_re, 48);
        p_im = wrapw<int64_t>(p_im, 48);
        const long long r = (p_re >> (xs - 1)), i = (p_im >> (xs - 1));
        e.x = wrapw<int32_t>((int32_t)(uint32_t)(uint64_t)((r >> 1) + (r & 1)), twd); // pr_rnd :176-199
        e.y = wrapw<int32_t>((int32_t)(uint32_t)(uint64_t)((i >> 1) + (i & 1)), twd);
    }
    out[k] = e;
}

hipError_t launch_twiddle_stage(const int2 *d_rom, int stage, int twd, int xser, int2 *d_out,
                                hipStream_t stream)
{
    const unsigned n = 1u << stage;
    const unsigned threads = 256;
    const int xs = xser ? 21 : 23;
    long long mathpi = 0;
    if (stage >= 11) // INTEGER(MATH_PI * 2.0**(13-ii-del_val)), del_val = 2 (NEW) / 0 (OLD)
        mathpi = llround(M_PI * ldexp(1.0, 13 - (stage - 11) - (xser ? 2 : 0)));
    hipLaunchKernelGGL(k_twiddle_stage, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, d_rom,
                       stage, twd, xs, mathpi, d_out);
    return hipGetLastError();
}

} // namespace intfft
