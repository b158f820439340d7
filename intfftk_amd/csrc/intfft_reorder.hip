// intfft_reorder.hip -- the reference's stream re-orderers as a standalone operator (intfft_reorder, include/intfft.h).
//
// Each buffer block of src/vhdl/buffers/ is a fixed permutation of one frame between two of the INTFFT_ORDER_* layouts:
//   inbuf_half_path.vhd:23-28        NATURAL      -> HALVES        (serial stream -> the two input lanes of int_fftNk)
//   outbuf_half_path.vhd:160-172     BITREV       -> BITREV_LANES  (output beats -> [lane0 frame ; lane1 frame])
//   int_bitrev_order.vhd:82-104,161-171  BITREV_LANES -> NATURAL   (bit_pair: keep the MSB, reverse the low NFFT-1 bits)
//   iobuf_flow_int2.vhd:18-40        NATURAL <-> beats of (even, odd) samples = NATURAL memory order (identity here)
// All four layouts are BIT PERMUTATIONS of the frame index, so any (from, to) pair is one: m_in bit i = m_out bit P[i].
// A workgroup moves one tile = the index bits {0..TB-1} of m_out (contiguous stores) united with the m_out bits that feed
// bits {0..TB-1} of m_in (contiguous loads): 2^U elements, TB <= U <= 2 TB, through LDS -- the classic tiled bit reversal,
// 256-byte runs on both sides for every pair of orders.  HBM-bound data movement, no arithmetic.
// multi-pass kernels: non-temporal loads measure 4-14 % faster here (the single-pass kernels gain 4-30 % from PLAIN loads): intfft_device.hpp
#define INTFFT_NT_LOADS 1
#include "../../include/intfft.h"
#include "intfft_pk16.hpp"

#include <algorithm>

namespace intfft {

typedef unsigned long long rv2u;                             // (re, im) of int32 containers, as one 8-byte word
typedef unsigned rv4u __attribute__((ext_vector_type(4))); // (re, im) of int64 containers

// Everything index-related is precomputed on the host as OR-masks, so that the kernel does no table look-ups in its
// loops (its first version indexed bit-position tables held in kernel-argument memory: two dependent scalar loads per
// bit per element group made it latency-bound at 1.3 TB/s).
//   element e = 256 i + tid of the LOAD enumeration  (consecutive e = consecutive m_in):
//       m_in = base_in | OR_k tid_k * t_in[k] | i_in[i],   LDS slot = OR_k tid_k * t_slot[k] | i_slot[i]
//   element e of the STORE enumeration (consecutive e = consecutive m_out, LDS slot = e):
//       m_out = base_out | OR_k tid_k * t_out[k] | i_out[i]
//   tile id bit k contributes r_in[k] / r_out[k] to base_in / base_out
struct ReorderArgs {
    int L, U;
    unsigned t_in[8], t_slot[8], t_out[8];
    unsigned i_in[16], i_slot[16], i_out[16];
    unsigned r_in[18], r_out[18];
};

template <typename E> __global__ __launch_bounds__(256) void k_reorder(const E *in, E *out, const ReorderArgs a, unsigned tiles_per_frame)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    E *lds = reinterpret_cast<E *>(lds_raw);
    const size_t frame = blockIdx.x / tiles_per_frame;
    const unsigned tile = blockIdx.x % tiles_per_frame;
    unsigned base_out = 0, base_in = 0; // contribution of the tile id (wave-uniform: scalar code)
#pragma unroll
    for (int k = 0; k < 18; ++k)
        if ((tile >> k) & 1u) {
            base_out |= a.r_out[k];
            base_in |= a.r_in[k];
        }
    const E *src = in + (frame << a.L) + base_in;
    E *dst = out + (frame << a.L) + base_out;
    const unsigned tid = threadIdx.x;
    unsigned t_in = 0, t_slot = 0, t_out = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned bit = 0u - ((tid >> k) & 1u);
        t_in |= bit & a.t_in[k];
        t_slot |= bit & a.t_slot[k];
        t_out |= bit & a.t_out[k];
    }
    const unsigned iters = a.U > 8 ? 1u << (a.U - 8) : 1u; // <= 16 (U <= 12)
    const bool active = a.U >= 8 || tid < (1u << a.U);
    // all loads of the thread are issued before the first LDS write: up to 16 independent requests in flight per lane
    E v[16];
#pragma unroll
    for (unsigned i = 0; i < 16; ++i)
        if (i < iters && active) v[i] = INTFFT_LD(src + (t_in | a.i_in[i]));
#pragma unroll
    for (unsigned i = 0; i < 16; ++i)
        if (i < iters && active) {
            const unsigned slot = t_slot | a.i_slot[i];
            lds[slot + (slot >> 5)] = v[i]; // + slot / 32: breaks the power-of-two bank stride of the transposing writes
        }
    __syncthreads();
#pragma unroll
    for (unsigned i = 0; i < 16; ++i)
        if (i < iters && active) {
            const unsigned e = tid + 256u * i;
            __builtin_nontemporal_store(lds[e + (e >> 5)], dst + (t_out | a.i_out[i]));
        }
}

// ---- the same mover with the 2-D scheme's multiplier between the cores fused in --------------------------
// TW = 1: V <- cmult(V, W_N^(k1 n2)) applied to the OUTPUT layout [k1][n2] (forward: [n2][k1] -> [k1][n2]);
// TW = 2: T = V * conj(W) through the re/im-swapped feed applied to the INPUT layout [k1][n2] (inverse: [k1][n2] -> [n2][k1]).
// A block owns one tile and walks the frames part, part + fsplit, ...: its 16 twiddles per thread are evaluated once.
struct TwArgs {
    int l2, mw, sh_a, sh_b, narrow, twd;
    int packed; // int16 containers, 16-bit data, twiddles of at most 16 bits (single-DSP regime): the packed multiplier of intfft_pk16.hpp
};
template <typename E> struct ElemIO;
template <> struct ElemIO<uint32_t> { // int16 containers
    typedef int32_t T;
    static __device__ __forceinline__ void get(uint32_t v, T &re, T &im) { re = (int16_t)(v & 0xFFFFu), im = (int16_t)(v >> 16); }
    static __device__ __forceinline__ uint32_t put(T re, T im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }
};
template <> struct ElemIO<rv2u> { // int32 containers
    typedef int32_t T;
    static __device__ __forceinline__ void get(rv2u v, T &re, T &im) { re = (int32_t)(uint32_t)v, im = (int32_t)(uint32_t)(v >> 32); }
    static __device__ __forceinline__ rv2u put(T re, T im) { return (rv2u)(uint32_t)re | ((rv2u)(uint32_t)im << 32); }
};
template <> struct ElemIO<rv4u> { // int64 containers
    typedef int64_t T;
    static __device__ __forceinline__ void get(rv4u v, T &re, T &im)
    {
        re = (int64_t)((uint64_t)v.x | ((uint64_t)v.y << 32));
        im = (int64_t)((uint64_t)v.z | ((uint64_t)v.w << 32));
    }
    static __device__ __forceinline__ rv4u put(T re, T im)
    {
        const rv4u r = {(unsigned)(uint64_t)re, (unsigned)((uint64_t)re >> 32), (unsigned)(uint64_t)im, (unsigned)((uint64_t)im >> 32)};
        return r;
    }
};

// PK (int16 containers only): the packed multiplier (TwArgs::packed) -- a template parameter, because both multipliers compiled
// into one kernel cost it 161 VGPRs (three waves per SIMD); the packed kernel alone takes half of that.
template <typename E, int TW, bool PK = false>
__global__ __launch_bounds__(256) void k_reorder_tw(const E *in, E *out, const ReorderArgs a, const TwArgs w, unsigned tiles, unsigned fsplit,
                                                    size_t nframes)
{
    typedef typename ElemIO<E>::T T;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    E *lds = reinterpret_cast<E *>(lds_raw);
    const unsigned tile = blockIdx.x % tiles, part = blockIdx.x / tiles;
    unsigned base_out = 0, base_in = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k)
        if ((tile >> k) & 1u) {
            base_out |= a.r_out[k];
            base_in |= a.r_in[k];
        }
    const unsigned tid = threadIdx.x;
    unsigned t_in = 0, t_slot = 0, t_out = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned bit = 0u - ((tid >> k) & 1u);
        t_in |= bit & a.t_in[k];
        t_slot |= bit & a.t_slot[k];
        t_out |= bit & a.t_out[k];
    }
    const unsigned iters = a.U > 8 ? 1u << (a.U - 8) : 1u;
    const bool active = a.U >= 8 || tid < (1u << a.U);
    // this thread's twiddles: positions of its loads (TW = 2) or of its stores (TW = 1) in the [k1][n2] layout
    int wr[16], wi[16];
    {
        double scale, mg;
        tw2d_consts(a.L, w.twd, scale, mg);
        const unsigned nmask = (1u << a.L) - 1u, m2 = (1u << w.l2) - 1u;
#pragma unroll
        for (unsigned i = 0; i < 16; ++i)
            if (i < iters) {
                const unsigned idx = TW == 1 ? (base_out | t_out | a.i_out[i]) : (base_in | t_in | a.i_in[i]);
                tw2d_eval(a.L, scale, mg, ((idx >> w.l2) * (idx & m2)) & nmask, wr[i], wi[i]);
                if constexpr (PK) { // forward: {Wa = (wr, -wi), Wb = (wi, wr)}; conjugate feed: {Wc = (wr, wi), Wd = (-wi, wr)}
                    const unsigned r16 = (unsigned)wr[i] & 0xFFFFu, i16 = (unsigned)wi[i] & 0xFFFFu, n16 = (unsigned)(-wi[i]) & 0xFFFFu;
                    const unsigned first = TW == 1 ? (r16 | (n16 << 16)) : (r16 | (i16 << 16));
                    const unsigned second = TW == 1 ? (i16 | (r16 << 16)) : (n16 | (r16 << 16));
                    wr[i] = (int)first, wi[i] = (int)second;
                }
                __builtin_amdgcn_sched_barrier(0); // one evaluation at a time: interleaved, the 16 double-precision chains set the kernel's VGPR count
            }
    }
    const Slice sl{w.twd - 1, w.twd, 0x05040100u, 0x07060302u};
    // one sample through the packed multiplier: Y = sum[t+14 : t-1] of the two dot products (mul2x<16>, two samples per call)
    auto pk_mul2 = [&](uint32_t &x0, uint32_t &x1, int i0, int i1) {
        uint32_t y0, y1;
        mul2x<16, false>(x0, x0, (uint32_t)wr[i0], (uint32_t)wi[i0], x1, x1, (uint32_t)wr[i1], (uint32_t)wi[i1], sl.off_y, sl.sel, y0, y1);
        x0 = y0, x1 = y1;
    };
    (void)pk_mul2;
    for (size_t f = part; f < nframes; f += fsplit) {
        const E *src = in + (f << a.L) + base_in;
        E *dst = out + (f << a.L) + base_out;
        // opaque copies of the loop-invariant thread offsets: otherwise LICM keeps 32 64-bit addresses (64 VGPRs) across the
        // frame loop; recomputing one costs an OR and the address add
        unsigned t_in_l = t_in, t_slot_l = t_slot, t_out_l = t_out;
        asm volatile("" : "+v"(t_in_l), "+v"(t_slot_l), "+v"(t_out_l));
        E v[16];
#pragma unroll
        for (unsigned i = 0; i < 16; ++i)
            if (i < iters && active) v[i] = INTFFT_LD(src + (t_in_l | a.i_in[i]));
        if constexpr (TW == 2 && PK) {
            if (active) { // T.re = dot(V, Wc), T.im = dot(V, Wd): the swapped feed needs no swap in this packing
#pragma unroll
                for (unsigned i = 0; i < 16; i += 2)
                    if (i < iters) {
                        uint32_t x0 = (uint32_t)v[i], x1 = (uint32_t)v[(i + 1) & 15];
                        pk_mul2(x0, x1, (int)i, (int)((i + 1) & 15));
                        v[i] = (E)x0;
                        if (i + 1 < iters) v[(i + 1) & 15] = (E)x1;
                    }
            }
        }
#pragma unroll
        for (unsigned i = 0; i < 16; ++i)
            if (i < iters && active) {
                if constexpr (TW == 2 && !PK) { // swapped feed: DI_RE <- im, DI_IM <- re; DO_RE -> im, DO_IM -> re (int_dit2_fly.vhd:304-322)
                    T re, im, ore, oim;
                    ElemIO<E>::get(v[i], re, im);
                    cmult(im, re, wr[i], wi[i], w.mw, w.sh_a, w.sh_b, w.narrow, ore, oim);
                    v[i] = ElemIO<E>::put(oim, ore);
                }
                const unsigned slot = t_slot_l | a.i_slot[i];
                lds[slot + (slot >> 5)] = v[i];
            }
        __syncthreads();
        if constexpr (TW == 1 && PK) {
            if (active) {
#pragma unroll
                for (unsigned i = 0; i < 16; i += 2)
                    if (i < iters) {
                        const unsigned e0 = tid + 256u * i, e1 = tid + 256u * ((i + 1) & 15);
                        uint32_t x0 = (uint32_t)lds[e0 + (e0 >> 5)], x1 = i + 1 < iters ? (uint32_t)lds[e1 + (e1 >> 5)] : 0u;
                        pk_mul2(x0, x1, (int)i, (int)((i + 1) & 15));
                        __builtin_nontemporal_store((E)x0, dst + (t_out_l | a.i_out[i]));
                        if (i + 1 < iters) __builtin_nontemporal_store((E)x1, dst + (t_out_l | a.i_out[(i + 1) & 15]));
                    }
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (unsigned i = 0; i < 16; ++i)
            if (i < iters && active) {
                const unsigned e = tid + 256u * i;
                E x = lds[e + (e >> 5)];
                if constexpr (TW == 1 && !PK) {
                    T re, im, ore, oim;
                    ElemIO<E>::get(x, re, im);
                    cmult(re, im, wr[i], wi[i], w.mw, w.sh_a, w.sh_b, w.narrow, ore, oim);
                    x = ElemIO<E>::put(ore, oim);
                }
                __builtin_nontemporal_store(x, dst + (t_out_l | a.i_out[i]));
            }
        __syncthreads(); // the next frame's LDS writes wait for these reads
    }
}

namespace {

// which memory-index bit supplies logical-index bit j in each layout (the maps of include/intfft.h)
int mem_bit_of_logical(int order, int L, int j)
{
    switch (order) {
    case INTFFT_ORDER_BITREV: return L - 1 - j;
    case INTFFT_ORDER_HALVES: return j == L - 1 ? 0 : j + 1;
    case INTFFT_ORDER_BITREV_LANES: return j == L - 1 ? L - 1 : L - 2 - j;
    default: return j;
    }
}

template <typename E> hipError_t launch(const ReorderArgs &a, const void *in, void *out, size_t batch, hipStream_t stream)
{
    const size_t lds = (((size_t)1 << a.U) + ((size_t)1 << a.U) / 32 + 1) * sizeof(E);
    if (lds > 48 * 1024) allow_max_lds(kptr(k_reorder<E>));
    const unsigned tiles = 1u << (a.L - a.U);
    const size_t blocks = batch * tiles;
    for (size_t b0 = 0; b0 < blocks; b0 += (size_t)1 << 30) { // grid.x limit
        const size_t nb = std::min<size_t>((size_t)1 << 30, blocks - b0);
        const size_t f0 = b0 / tiles; // 2^30 is a multiple of `tiles`
        hipLaunchKernelGGL(k_reorder<E>, dim3((unsigned)nb), dim3(256), lds, stream,
                           static_cast<const E *>(in) + (f0 << a.L), static_cast<E *>(out) + (f0 << a.L), a, tiles);
    }
    return hipGetLastError();
}

} // namespace
} // namespace intfft

using namespace intfft;

// any bit permutation of the frame index: m_in bit in_of_out[b] = m_out bit b (b = 0 .. L-1, L <= 24).  Used by
// intfft_reorder (pairs of INTFFT_ORDER_* layouts) and by the 2-D scheme plans (layout changes between the cores).
static ReorderArgs make_reorder_args(int L, const int *in_of_out)
{
    int out_of_in[24];
    for (int b = 0; b < L; ++b) out_of_in[in_of_out[b]] = b;
    const int TB = std::min(6, L); // 64-element runs: 256 B of int16 pairs
    bool in_tile[24] = {false};
    int out_pos[12], ld_pos[12], U = 0;
    for (int k = 0; k < TB; ++k) { // store enumeration: m_out bits 0..TB-1 first
        out_pos[U++] = k;
        in_tile[k] = true;
    }
    for (int k = 0; k < TB; ++k) { // then the m_out bits that feed m_in bits 0..TB-1
        const int b = out_of_in[k];
        if (!in_tile[b]) {
            out_pos[U++] = b;
            in_tile[b] = true;
        }
    }
    // load enumeration: the feeders of m_in bits 0..TB-1 in that order, then the remaining tile bits
    int nl = 0;
    bool used[24] = {false};
    for (int k = 0; k < TB; ++k) {
        ld_pos[nl++] = out_of_in[k];
        used[out_of_in[k]] = true;
    }
    for (int k = 0; k < U; ++k)
        if (!used[out_pos[k]]) ld_pos[nl++] = out_pos[k];
    ReorderArgs a{};
    a.L = L;
    a.U = U;
    auto st_of = [&](int ldk) { // store-enumeration bit of load-enumeration bit ldk
        for (int q = 0; q < U; ++q)
            if (out_pos[q] == ld_pos[ldk]) return q;
        return 0;
    };
    for (int k = 0; k < 8 && k < U; ++k) {
        a.t_in[k] = 1u << in_of_out[ld_pos[k]];
        a.t_slot[k] = 1u << st_of(k);
        a.t_out[k] = 1u << out_pos[k];
    }
    for (unsigned i = 0; i < 16; ++i)
        for (int k = 8; k < U; ++k)
            if ((i >> (k - 8)) & 1u) {
                a.i_in[i] |= 1u << in_of_out[ld_pos[k]];
                a.i_slot[i] |= 1u << st_of(k);
                a.i_out[i] |= 1u << out_pos[k];
            }
    // The remaining bits number the tiles.  Consecutive tile ids run concurrently, so the LOW tile-id bits should spread
    // BOTH sides over the memory channels: order the bits by min(position in m_out, position in m_in).
    int nr = 0;
    for (int key = 0; key < L; ++key)
        for (int b = 0; b < L; ++b)
            if (!in_tile[b] && std::min(b, in_of_out[b]) == key) {
                a.r_out[nr] = 1u << b;
                a.r_in[nr] = 1u << in_of_out[b];
                ++nr;
            }
    return a;
}

namespace intfft {
// One-bit rotations of the frame index -- BITREV <-> BITREV_LANES (outbuf_half_path.vhd:160-172: the two output lanes written out one
// after the other) and NATURAL <-> HALVES (inbuf_half_path.vhd:23-28) are the two directions of the same map -- need no tile: a thread
// that owns four consecutive samples of the interleaved side owns two pairs of consecutive samples of the split side.
//   SPLIT: out[h * N/2 + i] = in[2 i + h]      (m_out = (m_in & 1) << (L-1) | m_in >> 1)
//   !SPLIT: out[2 i + h] = in[h * N/2 + i]
// 16 / 32 / 64 bytes per thread on the interleaved side, 8 / 16 / 32-byte pieces on the split side, 512 B .. 2 KiB per wave instruction.
template <typename E> struct alignas(2 * sizeof(E)) Pair2 { E v[2]; };
template <typename E> struct alignas(4 * sizeof(E)) Quad4 { E v[4]; };
template <typename E, bool SPLIT>
__global__ __launch_bounds__(256) void k_rotate1(const E *__restrict__ in, E *__restrict__ out, int L, size_t nquads)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; // quad index over the whole batch
    if (t >= nquads) return;
    const size_t qpf = (size_t)1 << (L - 2);                 // quads per frame
    const size_t f = t >> (L - 2), q = t & (qpf - 1);
    const size_t base = f << L, half = (size_t)1 << (L - 1);
    if constexpr (SPLIT) {
        const Quad4<E> x = *reinterpret_cast<const Quad4<E> *>(in + base + 4 * q);
        const Pair2<E> ev = {{x.v[0], x.v[2]}}, od = {{x.v[1], x.v[3]}};
        *reinterpret_cast<Pair2<E> *>(out + base + 2 * q) = ev;
        *reinterpret_cast<Pair2<E> *>(out + base + half + 2 * q) = od;
    } else {
        const Pair2<E> ev = *reinterpret_cast<const Pair2<E> *>(in + base + 2 * q), od = *reinterpret_cast<const Pair2<E> *>(in + base + half + 2 * q);
        const Quad4<E> x = {{ev.v[0], od.v[0], ev.v[1], od.v[1]}};
        *reinterpret_cast<Quad4<E> *>(out + base + 4 * q) = x;
    }
}
template <typename E> static hipError_t launch_rotate1(bool split, int L, const void *in, void *out, size_t batch, hipStream_t stream)
{
    const size_t nquads = batch << (L - 2), per = (size_t)1 << 30; // grid.x limit: 2^30 blocks of 256 quads per launch
    for (size_t b0 = 0; b0 < (nquads + 255) / 256; b0 += per) {
        const size_t nb = std::min(per, (nquads + 255) / 256 - b0), off = b0 * 256 * 4; // samples before this launch (a multiple of the frame)
        if (split) hipLaunchKernelGGL((k_rotate1<E, true>), dim3((unsigned)nb), dim3(256), 0, stream, static_cast<const E *>(in) + off, static_cast<E *>(out) + off, L, nquads - b0 * 256);
        else hipLaunchKernelGGL((k_rotate1<E, false>), dim3((unsigned)nb), dim3(256), 0, stream, static_cast<const E *>(in) + off, static_cast<E *>(out) + off, L, nquads - b0 * 256);
    }
    return hipGetLastError();
}
} // namespace intfft

namespace intfft {
// USE_FLY = 0 (the bypass mux of int_fftNk.vhd:260-277): no butterfly touches the data; what remains of the arithmetic is the width handling
// of the first stage -- the input wrapped to DATA_WIDTH bits (sign-extended; zero-extended when FORMAT = 1 widens the word, as the RTL's
// bypass does) in the output container.  (The data movement of the commutators is a bit permutation: intfft_plan.hip, lanes_mode 3.)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_convert(const TI *__restrict__ in, TO *__restrict__ out, size_t n, int dw, int zext)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int sh = 64 - dw;
    auto cv = [&](TI x) -> TO {
        const long long v = (long long)x;
        const unsigned long long u = (unsigned long long)v << sh; // the low dw bits at the top of the word
        return zext ? (TO)(long long)(u >> sh) : (TO)((long long)u >> sh);
    };
    if (i + 4 <= n) {
        struct alignas(4 * sizeof(TI)) QI { TI v[4]; };
        struct alignas(4 * sizeof(TO)) QO { TO v[4]; };
        const QI x = *reinterpret_cast<const QI *>(in + i);
        QO y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y.v[k] = cv(x.v[k]);
        *reinterpret_cast<QO *>(out + i) = y;
    } else {
        for (size_t k = i; k < n; ++k) out[k] = cv(in[k]);
    }
}
template <typename TI, typename TO> static hipError_t launch_convert_t(const void *in, void *out, size_t n, int dw, int zext, hipStream_t stream)
{
    const size_t per = (size_t)1 << 30; // scalars per launch (a multiple of 4 x 256)
    for (size_t o = 0; o < n; o += per) {
        const size_t m = std::min(per, n - o);
        hipLaunchKernelGGL((k_convert<TI, TO>), dim3((unsigned)((m + 1023) / 1024)), dim3(256), 0, stream, static_cast<const TI *>(in) + o,
                           static_cast<TO *>(out) + o, m, dw, zext);
    }
    return hipGetLastError();
}
// n scalars (2 per complex sample) of in_cb-byte containers -> out_cb-byte containers (2 / 4 / 8, out_cb >= in_cb)
hipError_t launch_convert(int in_cb, int out_cb, int dw, int zext, const void *in, void *out, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    switch (in_cb * 16 + out_cb) {
    case 2 * 16 + 2: return launch_convert_t<short, short>(in, out, n, dw, zext, stream);
    case 2 * 16 + 4: return launch_convert_t<short, int>(in, out, n, dw, zext, stream);
    case 2 * 16 + 8: return launch_convert_t<short, long long>(in, out, n, dw, zext, stream);
    case 4 * 16 + 4: return launch_convert_t<int, int>(in, out, n, dw, zext, stream);
    case 4 * 16 + 8: return launch_convert_t<int, long long>(in, out, n, dw, zext, stream);
    case 8 * 16 + 8: return launch_convert_t<long long, long long>(in, out, n, dw, zext, stream);
    default: return hipErrorInvalidValue;
    }
}
} // namespace intfft

hipError_t intfft::launch_bitperm(int L, int container_bytes, const int *in_of_out, const void *d_in, void *d_out, size_t batch,
                                  hipStream_t stream)
{
    if (L >= 3 && !diag_env("INTFFT_NO_ROTATE1")) { // one-bit rotations of the index: a streaming kernel, no tile
        bool up = true, down = true; // up: m_in bit (b+1) mod L = m_out bit b (the split direction); down: the inverse
        for (int b = 0; b < L; ++b) {
            up = up && in_of_out[b] == (b + 1) % L;
            down = down && in_of_out[b] == (b + L - 1) % L;
        }
        if (up || down) {
            switch (container_bytes) {
            case 2: return launch_rotate1<uint32_t>(up, L, d_in, d_out, batch, stream);
            case 4: return launch_rotate1<rv2u>(up, L, d_in, d_out, batch, stream);
            default: return launch_rotate1<rv4u>(up, L, d_in, d_out, batch, stream);
            }
        }
    }
    const ReorderArgs a = make_reorder_args(L, in_of_out);
    switch (container_bytes) {
    case 2: return launch<uint32_t>(a, d_in, d_out, batch, stream);
    case 4: return launch<rv2u>(a, d_in, d_out, batch, stream);
    default: return launch<rv4u>(a, d_in, d_out, batch, stream);
    }
}

template <typename E>
static hipError_t launch_tw(const ReorderArgs &a, const TwArgs &w, int conj, const void *in, void *out, size_t batch, hipStream_t stream)
{
    const size_t lds = (((size_t)1 << a.U) + ((size_t)1 << a.U) / 32 + 1) * sizeof(E);
    const unsigned tiles = 1u << (a.L - a.U);
    unsigned fsplit = 1; // blocks per tile: enough blocks to fill the chip, as many frames per block as that leaves
    const size_t want = (size_t)device_cus() * 8;
    while ((size_t)tiles * fsplit < want && (size_t)fsplit * 2 <= batch) fsplit *= 2;
    auto go = [&](auto kernel) {
        if (lds > 48 * 1024) allow_max_lds(kptr(kernel));
        hipLaunchKernelGGL(kernel, dim3(tiles * fsplit), dim3(256), lds, stream, static_cast<const E *>(in), static_cast<E *>(out), a, w,
                           tiles, fsplit, batch);
    };
    if constexpr (sizeof(E) == 4) {
        if (w.packed) {
            if (conj) go(k_reorder_tw<E, 2, true>);
            else go(k_reorder_tw<E, 1, true>);
            return hipGetLastError();
        }
    }
    if (conj) go(k_reorder_tw<E, 2>);
    else go(k_reorder_tw<E, 1>);
    return hipGetLastError();
}

// the layout change between the cores of a 2-D scheme plan with the multiplier fused in: conj = 0 multiplies in the OUTPUT
// layout (which must be [k1][n2]), conj = 1 in the INPUT layout
hipError_t intfft::launch_bitperm_tw(int L, int container_bytes, const int *in_of_out, int l2, int mw, int sh_a, int sh_b, int narrow,
                                     int twd, int conj, const void *d_in, void *d_out, size_t batch, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    const ReorderArgs a = make_reorder_args(L, in_of_out);
    static const int no_packed = diag_env("INTFFT_2D_NO_PACKED_TW") ? 1 : 0; // A/B: the general multiplier on int16 containers too
    const TwArgs w{l2, mw, sh_a, sh_b, narrow, twd, (container_bytes == 2 && mw == 16 && twd <= 16 && sh_a == 0 && sh_b == twd - 1 && !no_packed) ? 1 : 0};
    switch (container_bytes) {
    case 2: return launch_tw<uint32_t>(a, w, conj, d_in, d_out, batch, stream);
    case 4: return launch_tw<rv2u>(a, w, conj, d_in, d_out, batch, stream);
    default: return launch_tw<rv4u>(a, w, conj, d_in, d_out, batch, stream);
    }
}

int intfft::order_mem_bit(int order, int L, int j) { return mem_bit_of_logical(order, L, j); }

extern "C" int intfft_reorder(int log2n, int container_bytes, int from_order, int to_order, const void *d_in, void *d_out,
                              size_t batch, int hip_device, void *hip_stream)
{
    if (log2n < 3 || log2n > 24) return INTFFT_ERR_INVALID;
    if (container_bytes != 2 && container_bytes != 4 && container_bytes != 8) return INTFFT_ERR_INVALID;
    if (from_order < 0 || from_order > 3 || to_order < 0 || to_order > 3) return INTFFT_ERR_INVALID;
    if (batch && (!d_in || !d_out)) return INTFFT_ERR_NULL;
    if (batch == 0) return INTFFT_OK;
    const int L = log2n;
    {   // a permutation cannot run in place
        const size_t bytes = batch * ((size_t)2 << L) * (size_t)container_bytes;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(d_in), b0 = reinterpret_cast<uintptr_t>(d_out);
        if (a0 < b0 + bytes && b0 < a0 + bytes) return INTFFT_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || hip_device < 0 || hip_device >= ndev) return INTFFT_ERR_NO_DEVICE;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(hip_device) != hipSuccess) return INTFFT_ERR_NO_DEVICE;
    int in_of_out[24]; // logical bit j: m_in bit mem_bit(from, j) = m_out bit mem_bit(to, j)
    for (int j = 0; j < L; ++j) in_of_out[mem_bit_of_logical(to_order, L, j)] = mem_bit_of_logical(from_order, L, j);
    const hipError_t e = launch_bitperm(L, container_bytes, in_of_out, d_in, d_out, batch, reinterpret_cast<hipStream_t>(hip_stream));
    if (prev >= 0) (void)hipSetDevice(prev);
    return (int)e;
}
