// intfft_pass16.hip -- packed-int16 LDS pass kernel: every scaled (FORMAT = 0) configuration with
// DATA_WIDTH = 16 and TWDL_WIDTH <= 16 that the N = 1024 wave kernel does not serve: any length
// 8 .. 2^20 (multi-pass above 2^14), int_fftNk / int_ifftNk / the FFT->IFFT pair, truncate or round,
// all four I/O orders.  Same tiling and planner as k_pass<T> (intfft_generic.hip); the difference is
// the arithmetic: samples stay packed (re | im << 16) in LDS and every butterfly is
//   2 x v_pk_ashrrev_i16, v_pk_add_u16, v_pk_sub_i16, 2 x v_dot2_i32_i16, 2 x v_bfe_i32, v_perm_b32
// (int_dif2_fly.vhd:144-373, int_dit2_fly.vhd:142-325, int_cmult_dsp48.vhd:184-225 "sngl" regime),
// issued two butterflies at a time from the hazard-safe asm block of intfft_fast1024.hip.
//
// Twiddles come from a packed table built at plan time (two u32 per entry):
//   DIF:  Wa = (wr, -wi), Wb = (wi,  wr)   re = dot(D, Wa),  im = dot(D, Wb)        Y = D * W
//   DIT:  Wc = (wr,  wi), Wd = (-wi, wr)   re = dot(B, Wc),  im = dot(B, Wd)        T = B * conj(W)
// (the DIT multiplier is fed re/im-swapped, int_dit2_fly.vhd:304-322; algebraically that is the
// conj(W) product, and both dot products are exact in int32 so the operand order is immaterial).
#include "intfft_internal.hpp"

namespace intfft {

using u32 = uint32_t;
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s p16_v2s(u32 x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ u32 p16_u32(v2s x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ unsigned pad16(unsigned e) { return e + (e >> 5); }

// two complex multiplies, exact extraction of sum[off+15 : off] (floor, 16-bit wrap)
__device__ __forceinline__ void p16_mul2(u32 d0, u32 wa0, u32 wb0, u32 d1, u32 wa1, u32 wb1, int off, u32 &y0,
                                         u32 &y1)
{
    u32 r0, i0, r1, i1;
    const u32 sel = 0x05040100u;
    asm("v_dot2_i32_i16 %[r0], %[d0], %[wa0], 0\n\t"
        "v_dot2_i32_i16 %[i0], %[d0], %[wb0], 0\n\t"
        "v_dot2_i32_i16 %[r1], %[d1], %[wa1], 0\n\t"
        "v_dot2_i32_i16 %[i1], %[d1], %[wb1], 0\n\t"
        "v_bfe_i32 %[y0], %[r0], %[off], 16\n\t"
        "v_bfe_i32 %[r0], %[i0], %[off], 16\n\t"
        "v_bfe_i32 %[y1], %[r1], %[off], 16\n\t"
        "v_bfe_i32 %[i0], %[i1], %[off], 16\n\t"
        "v_perm_b32 %[y0], %[r0], %[y0], %[sel]\n\t"
        "v_perm_b32 %[y1], %[i0], %[y1], %[sel]"
        : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
        : [d0] "v"(d0), [wa0] "v"(wa0), [wb0] "v"(wb0), [d1] "v"(d1), [wa1] "v"(wa1), [wb1] "v"(wb1), [off] "s"(off),
          [sel] "s"(sel));
}

// S, D (DIF) or X, Y (DIT) with the two scaled variants
__device__ __forceinline__ void p16_addsub(u32 a, u32 b, bool round, u32 &s, u32 &d)
{
    const v2s A = p16_v2s(a), B = p16_v2s(b);
    if (!round) { // (A >> 1) +/- (B >> 1): LSB dropped before the add (int_dif2_fly.vhd:151-154)
        const v2s A1 = A >> (short)1, B1 = B >> (short)1;
        s = p16_u32(A1 + B1);
        d = p16_u32(A1 - B1);
    } else { // rhu2 of the exact sums, wrapped to 16 bits (:173-218)
        const v2s T = (A ^ B) >> (short)1; // rhu2(A + B) = (A | B) - T, rhu2(A - B) = rhu2(A + B) - B (derivation: sumdiff, intfft_pk16.hpp)
        const v2s S = (A | B) - T;
        s = p16_u32(S);
        d = p16_u32(S - B);
    }
}

// x >= 0 ? -x : ~x on ONE 16-bit half (HI selects it), the other half untouched
template <bool HI> __device__ __forceinline__ u32 p16_negq(u32 v)
{
    const u32 nx = v ^ (HI ? 0xFFFF0000u : 0x0000FFFFu);
    const short c = (short)((nx >> (HI ? 31 : 15)) & 1u); // 1 iff ~x < 0 iff x >= 0
    const v2s add = HI ? v2s{0, c} : v2s{c, 0};
    return p16_u32(p16_v2s(nx) + add);
}

// one complex multiply (groups with a single butterfly per stage): the s_nop keeps the DOT -> VALU
// distance that p16_mul2 gets from interleaving
__device__ __forceinline__ void p16_mul1(u32 d0, u32 wa0, u32 wb0, int off, u32 &y0)
{
    u32 r0, i0;
    const u32 sel = 0x05040100u;
    asm("v_dot2_i32_i16 %[r0], %[d0], %[wa0], 0\n\t"
        "v_dot2_i32_i16 %[i0], %[d0], %[wb0], 0\n\t"
        "s_nop 2\n\t"
        "v_bfe_i32 %[y0], %[r0], %[off], 16\n\t"
        "s_nop 0\n\t"
        "v_bfe_i32 %[r0], %[i0], %[off], 16\n\t"
        "v_perm_b32 %[y0], %[r0], %[y0], %[sel]"
        : [y0] "=&v"(y0), [r0] "=&v"(r0), [i0] "=&v"(i0)
        : [d0] "v"(d0), [wa0] "v"(wa0), [wb0] "v"(wb0), [off] "s"(off), [sel] "s"(sel));
}

// multiplier-free butterflies of STAGE 0 / 1 (k = twiddle counter of the pair)
template <int KIND> __device__ __forceinline__ void p16_fly_triv(u32 &a, u32 &b, int s, unsigned k, bool round)
{
    if (KIND == KIND_DIF) {
        u32 x, d;
        p16_addsub(a, b, round, x, d);
        a = x;
        b = (s == 1 && (k & 1u)) ? p16_negq<true>(__builtin_amdgcn_alignbit(d, d, 16)) : d;
    } else {
        const u32 t = (s == 1 && (k & 1u)) ? p16_negq<false>(__builtin_amdgcn_alignbit(b, b, 16)) : b;
        p16_addsub(a, t, round, a, b);
    }
}

// 2^R-point sub-transform in registers: element r of the group lives at lds[pad16(e + (r << lb_lo))]
// and carries index bits s_lo .. s_lo+R-1 = r.  Stage s_lo+i pairs r-bit i and uses twiddle
// kb + ((r mod 2^i) << s_lo) of its table (kb = the group's index bits below s_lo).
template <int KIND, int R>
__device__ __forceinline__ void round16(u32 *lds, unsigned e, int lb_lo, int s_lo, unsigned kb, const uint2 *tw,
                                        bool round, int tsh)
{
    constexpr int P = 1 << R;
    u32 v[P];
#pragma unroll
    for (int r = 0; r < P; ++r) v[r] = lds[pad16(e + ((unsigned)r << lb_lo))];
#pragma unroll
    for (int ii = 0; ii < R; ++ii) {
        const int i = KIND == KIND_DIF ? R - 1 - ii : ii;
        const int s = s_lo + i;
        const int h = 1 << i;
        const uint2 *tws = tw + ((1u << s) - 1u) + kb; // stage table at offset 2^s - 1
        if (s >= 2) {
            // general butterflies, two at a time where the group has two
            if (P >= 4) {
#pragma unroll
                for (int pr = 0; pr < P / 2; pr += 2) {
                    // pair index -> r with bit i clear
                    const int r0 = ((pr >> i) << (i + 1)) | (pr & (h - 1));
                    const int r1 = (((pr + 1) >> i) << (i + 1)) | ((pr + 1) & (h - 1));
                    const uint2 w0 = tws[(unsigned)(r0 & (h - 1)) << s_lo], w1 = tws[(unsigned)(r1 & (h - 1)) << s_lo];
                    if (KIND == KIND_DIF) {
                        u32 d0, d1;
                        p16_addsub(v[r0], v[r0 + h], round, v[r0], d0);
                        p16_addsub(v[r1], v[r1 + h], round, v[r1], d1);
                        p16_mul2(d0, w0.x, w0.y, d1, w1.x, w1.y, tsh, v[r0 + h], v[r1 + h]);
                    } else {
                        u32 t0, t1;
                        p16_mul2(v[r0 + h], w0.x, w0.y, v[r1 + h], w1.x, w1.y, tsh, t0, t1);
                        p16_addsub(v[r0], t0, round, v[r0], v[r0 + h]);
                        p16_addsub(v[r1], t1, round, v[r1], v[r1 + h]);
                    }
                }
            } else {
                const uint2 w0 = tws[0];
                if (KIND == KIND_DIF) {
                    u32 d0;
                    p16_addsub(v[0], v[1], round, v[0], d0);
                    p16_mul1(d0, w0.x, w0.y, tsh, v[1]);
                } else {
                    u32 t0;
                    p16_mul1(v[1], w0.x, w0.y, tsh, t0);
                    p16_addsub(v[0], t0, round, v[0], v[1]);
                }
            }
        } else {
#pragma unroll
            for (int pr = 0; pr < P / 2; ++pr) {
                const int r0 = ((pr >> i) << (i + 1)) | (pr & (h - 1));
                p16_fly_triv<KIND>(v[r0], v[r0 + h], s, kb + ((unsigned)(r0 & (h - 1)) << s_lo), round);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < P; ++r) lds[pad16(e + ((unsigned)r << lb_lo))] = v[r];
}

__global__ __launch_bounds__(1024) void k_pass16(const PassArgs a, const void *in, void *out,
                                                         const uint2 *__restrict__ twf,
                                                         const uint2 *__restrict__ twi, size_t nframes, int tsh)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    u32 *lds = reinterpret_cast<u32 *>(smem16);

    const int L = a.L, U = a.U;
    const unsigned tiles = 1u << (L - U);
    const unsigned tile = blockIdx.x % tiles;
    const size_t f0 = (size_t)(blockIdx.x / tiles) * (size_t)a.fpb;
    const unsigned nf = (unsigned)min((size_t)a.fpb, nframes - f0);
    const size_t N = (size_t)1 << L;

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
    auto spread = [&](unsigned u) -> unsigned { return tile_bits | ((u & m0) << a.pos0) | ((u >> a.len0) << a.pos1); };
    auto swap_runs = [&](unsigned v) -> unsigned {
        const unsigned m1 = (1u << a.len1) - 1u;
        return ((v & m1) << a.len0) | (v >> a.len1);
    };
    const unsigned tile_n = 1u << U;
    const unsigned total = nf << U;
    const u32 *uin = reinterpret_cast<const u32 *>(in); // int16 (re, im) pairs == packed words
    u32 *uout = reinterpret_cast<u32 *>(out);

    // ---- load ----
    for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
        const unsigned f = i >> U, v = i & (tile_n - 1u);
        unsigned u = a.ld_swap ? swap_runs(v) : v;
        const unsigned j = spread(u);
        size_t m = j;
        if (a.in_mode == IO_USER) {
            if (a.ld_memorder) { // v is the memory index; the tile owns every bit, so u == core index
                const unsigned logical = order_from_mem(a.in_order, L, v);
                u = a.in_rev ? brev_l(logical, L) : logical;
                m = v;
            } else {
                m = order_to_mem(a.in_order, L, a.in_rev ? brev_l(j, L) : j);
            }
        }
        lds[pad16((f << U) + u)] = uin[(f0 + f) * N + m];
    }
    __syncthreads();

    // ---- stages: up to four consecutive stages per LDS round trip, evaluated in registers ----
    // Stages of one pass act on consecutive tile-local bits (descending for DIF, ascending for DIT),
    // so a run of R <= 4 stages is a 2^R-point sub-transform per thread (round16<>).
    int si = 0;
    while (si < a.nstages) {
        const StageDesc st = a.st[si];
        int R = 1;
        while (R < 4 && si + R < a.nstages && a.st[si + R].kind == st.kind &&
               a.st[si + R].lb == st.lb + (st.kind == KIND_DIF ? -R : R))
            ++R;
        const int lb_lo = st.kind == KIND_DIF ? st.lb - (R - 1) : st.lb;
        const int s_lo = st.kind == KIND_DIF ? st.s - (R - 1) : st.s;
        const bool round = st.rnd == RND_ROUND;
        const uint2 *tw = st.kind == KIND_DIF ? twf : twi;
        const unsigned ngroups = nf << (U - R);
        const unsigned lowm = (1u << lb_lo) - 1u;
        for (unsigned g = threadIdx.x; g < ngroups; g += blockDim.x) {
            const unsigned f = g >> (U - R), gg = g & ((tile_n >> R) - 1u);
            const unsigned u0 = ((gg >> lb_lo) << (lb_lo + R)) | (gg & lowm);
            const unsigned kb = spread(u0) & ((1u << s_lo) - 1u);
            u32 *base = lds;
            const unsigned e = (f << U) + u0;
            if (st.kind == KIND_DIF) {
                switch (R) {
                case 4: round16<KIND_DIF, 4>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                case 3: round16<KIND_DIF, 3>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                case 2: round16<KIND_DIF, 2>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                default: round16<KIND_DIF, 1>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                }
            } else {
                switch (R) {
                case 4: round16<KIND_DIT, 4>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                case 3: round16<KIND_DIT, 3>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                case 2: round16<KIND_DIT, 2>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                default: round16<KIND_DIT, 1>(base, e, lb_lo, s_lo, kb, tw, round, tsh); break;
                }
            }
        }
        __syncthreads();
        si += R;
    }

    // ---- store ----
    for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
        const unsigned f = i >> U, v = i & (tile_n - 1u);
        unsigned u = a.st_swap ? swap_runs(v) : v;
        const unsigned j = spread(u);
        size_t m = j;
        if (a.out_mode == IO_USER) {
            if (a.st_memorder) {
                const unsigned logical = order_from_mem(a.out_order, L, v);
                u = a.out_rev ? brev_l(logical, L) : logical;
                m = v;
            } else {
                m = order_to_mem(a.out_order, L, a.out_rev ? brev_l(j, L) : j);
            }
        }
        uout[(f0 + f) * N + m] = lds[pad16((f << U) + u)];
    }
}

bool pass16_supported(int data_width, int twdl_width, int format, int use_fly)
{
    return data_width == 16 && twdl_width >= 4 && twdl_width <= 16 && format == 0 && use_fly == 1;
}

size_t pass16_lds_bytes(const PassArgs &a)
{
    const size_t elems = (size_t)a.fpb << a.U;
    return (elems + (elems >> 5) + 1) * sizeof(u32);
}

const char *pass16_kernel_name() { return "k_pass16"; }

hipError_t launch_pass16(const PassArgs &a, const void *in, void *out, const uint2 *twf, const uint2 *twi,
                         size_t nframes, int twd, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    const size_t groups = (nframes + (size_t)a.fpb - 1) / (size_t)a.fpb;
    const size_t blocks = groups << (a.L - a.U);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    allow_max_lds(kptr(&k_pass16));
    hipLaunchKernelGGL(k_pass16, dim3((unsigned)blocks), dim3(pass_threads(a)), pass16_lds_bytes(a), stream, a, in, out,
                       twf, twi, nframes, twd - 1);
    return hipGetLastError();
}

// packs the int2 (re, im) twiddle buffer into the two dot-product operand forms
__global__ void k_pack_twiddles16(const int2 *__restrict__ tw, size_t n, uint2 *__restrict__ f, uint2 *__restrict__ i)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int2 w = tw[k];
    const u32 wr = (u32)w.x & 0xFFFFu, wi = (u32)w.y & 0xFFFFu, nwi = (u32)(-w.y) & 0xFFFFu;
    f[k] = make_uint2(wr | (nwi << 16), wi | (wr << 16)); // Wa = (wr, -wi), Wb = (wi, wr)
    i[k] = make_uint2(wr | (wi << 16), nwi | (wr << 16)); // Wc = (wr,  wi), Wd = (-wi, wr)
}

hipError_t launch_pack_twiddles16(const int2 *tw, size_t n, uint2 *f, uint2 *i, hipStream_t stream)
{
    hipLaunchKernelGGL(k_pack_twiddles16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tw, n, f, i);
    return hipGetLastError();
}

} // namespace intfft
