// intfft_fast1024u.hip -- UNSCALED (full bit growth) wave kernel for N = 64 .. 1024 (template parameter L = log2 N):
// int_fftNk with NFFT = 6..10, DATA_WIDTH = 16, TWDL_WIDTH <= 16, FORMAT = 1, natural in -> natural out
// (the reference testbench's primary mode: fft_signle_test.vhd:93-112 "UNSCALED", shipped with NFFT = 7).
// 16-bit samples in int16 containers, (16 + L)-bit results in int32 containers (12 B per complex sample of HBM
// traffic).  The description below is for N = 1024; shorter frames share a wave (2^(10-L) frames per 1024-sample
// chunk, the stages of the frame-number bits skipped) as in intfft_fast1024.hip, with the store mapping of
// lane_bit_u<L>() (intfft_u32.hpp).
//
// Same wave-per-frame choreography as intfft_fast1024.hip (stages 9..6 in registers, two lane swaps for
// stages 5 and 4, one wave-private LDS transpose, stages 3..0 in registers, bit reversal folded into the
// transpose), but on unpacked int32 (re, im) registers, because the width grows by one bit per stage
// (int_fftNk.vhd:187-207: DTW = 16 + ii, outputs 17 + ii bits):
//   S = A + B, D = A - B                      exact, one bit of growth (int_dif2_fly.vhd:222-240)
//   re = D.re*wr - D.im*wi, im = D.re*wi + D.im*wr   exact 64-bit sums: 2 x v_mad_i64_i32 each
//                                             (tools/valubench.hip: same issue rate as any VOP3 op)
//   Y = wrap_wo(sum >> (t-1))                 v_alignbit_b32 (+ v_bfe_i32 for the wo-bit wrap)
// The wo-bit wrap only matters when a rotated difference leaves the wo-bit range, i.e. for inputs
// beyond one guard bit: frames that pass the per-wave guard-bit vote (|re|, |im| <= 2^14: the complex
// magnitude is <= 2^14.5 * 2^k after k stages, below the 2^(15+k) range, and the per-stage floor adds
// < 1) skip the wrap; every other frame takes the exact path.  Single-DSP multiplier regime
// (w <= 26 < 28: int_cmult_dsp48.vhd:184-225).
#include "intfft_u32.hpp"

namespace intfft {

// NAT (round 4): the instantiation for int_fftNk's own beat orders, selected by `native` (bit 0: HALVES in, bit 1: BITREV out; N >= 128) --
//   HALVES  a beat (x[i], x[i + N/2]) is the register pair (j0, j0 | 2^(L-7)) of one lane: eight 8-byte loads, 512 contiguous bytes per wave instruction
//   BITREV  memory index = core position: after the last stage a lane holds 16 CONSECUTIVE positions (reg = a3..a0); the chunk goes through the wave's
//           idle LDS tile (rows of 16 samples, 36 dwords apart) and leaves in memory order, 1 KiB per wave instruction
// The natural-order instantiation (NAT = false) carries none of it.
template <int L, bool FAST_OK, bool NAT = false>
__global__ __launch_bounds__(256) void k_fft1024_u32(const u32 *in, int2 *out, const int2 *__restrict__ twt, const UConsts c,
                                                     size_t nframes_user, int sh, int native)
{
    static_assert(!NAT || L >= 7, "native beat orders: N >= 128");
    const bool halves = NAT && (native & 1), bitrev = NAT && (native & 2);
    constexpr int FP = 1 << (10 - L);                    // frames per 1024-sample chunk (intfft_fast1024.hip)
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * 2 * 64 * ROWU];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *lds = lds_all + wv * 2 * 64 * ROWU;

    int w9r[8] = {}, w9i[8] = {}, w8r[4] = {}, w8i[4] = {}, w7r[2] = {}, w7i[2] = {}, w6r = 0, w6i = 0, w5r, w5i, w4r, w4i;
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int2 w = twt[511 + 64 * j + lane];
            w9r[j] = w.x;
            w9i[j] = w.y;
        }
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int2 w = twt[255 + 64 * j + lane];
            w8r[j] = w.x;
            w8i[j] = w.y;
        }
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int2 w = twt[127 + 64 * j + lane];
            w7r[j] = w.x;
            w7i[j] = w.y;
        }
    }
    {
        int2 w;
        if constexpr (L >= 7) {
            w = twt[63 + lane];
            w6r = w.x;
            w6i = w.y;
        }
        w = twt[31 + (lane & 31)];
        w5r = w.x;
        w5i = w.y;
        w = twt[15 + (lane & 15)];
        w4r = w.x;
        w4i = w.y;
    }
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    u32 *wr_base = lds + ROWU * ((t5 << lane_bit_u<L>(9)) + (t4 << lane_bit_u<L>(8))) + (lane & 15);
    int lane_off = 0, lane_frame = 0; // N < 1024: see intfft_fast1024.hip
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
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROWU * lane);

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const u32 *src = in + f * 1024 + lane;
        u32 raw[16];
        if (NAT && halves) {
            // pair index of (register pair jj, lane): logical position P = 64 j0 + lane with bit L-1 clear -> frame P >> L, beat P mod N/2
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *src2 = reinterpret_cast<const v2u *>(in + f * 1024) + lane;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HB = 1 << (L >= 7 ? L - 7 : 0);                 // register bit that carries a(L-1)
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB);
                const int p0 = 64 * j0;                                       // register part of P
                const int pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                v2u w = {0u, 0u};
                if (!(L < 10) || f * FP + (size_t)(p0 >> L) < nframes_user) w = INTFFT_LD(src2 + pair);
                raw[j0] = w.x;
                raw[j0 | HB] = w.y;
            }
        } else if (L < 10 && (f + 1) * FP > nframes_user) { // partial last chunk: absent frames read as 0
#pragma unroll
            for (int j = 0; j < 16; ++j)
                raw[j] = f * FP + (size_t)((64 * j + lane) >> L) < nframes_user ? src[64 * j] : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) raw[j] = INTFFT_LD(src + 64 * j);
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
        if (FAST_OK && fast) utransform<L, false>(re, im, w9r, w9i, w8r, w8i, w7r, w7i, w6r, w6i, w5r, w5i, w4r, w4i, c, sh, wr_base, rd_base);
        else utransform<L, true>(re, im, w9r, w9i, w8r, w8i, w7r, w7i, w6r, w6i, w5r, w5i, w4r, w4i, c, sh, wr_base, rd_base);
        if (NAT && bitrev) {
            // position of (lane, reg r) = A(lane) | r with A = the index bits a9..a4 the lane carries (lane_bit_u<L>): 16 consecutive samples per lane
            typedef int v4i __attribute__((ext_vector_type(4)));
            int A = 0;
#pragma unroll
            for (int k = 4; k < 10; ++k) A |= ((lane >> lane_bit_u<L>(k)) & 1) << k;
            wave_lds_fence();
            u32 *row = lds + 36 * (A >> 4); // 16 samples = 32 dwords per row, 4 dwords of pad
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const v4i y = {re[2 * q], im[2 * q], re[2 * q + 1], im[2 * q + 1]};
                *reinterpret_cast<v4i *>(row + 4 * q) = y;
            }
            wave_lds_fence();
            v4i *dst4 = reinterpret_cast<v4i *>(out + f * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 64 * i + lane; // 16-byte piece = samples 2 e, 2 e + 1 of the chunk
                if (L < 10 && f * FP + (size_t)((2 * e) >> L) >= nframes_user) continue;
                __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 36 * (e >> 3) + 4 * (e & 7)), dst4 + e);
            }
            wave_lds_fence();
        } else if constexpr (L < 10) {
            // one lane swap per plane: reg bit 3 = a(L-1); registers {q, q+8} are two consecutive outputs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uswap32(re[r], re[r + 8]);
                uswap32(im[r], im[r + 8]);
            }
            typedef int v4i __attribute__((ext_vector_type(4)));
            int2 *dst = out + f * 1024 + lane_off;
            if constexpr (L == 6) {
                // N = 64: two lanes share a run, a store instruction would write thirty-two 32-byte runs -- the finished chunk (8 KiB) goes
                // through the wave's idle LDS tile in memory order and leaves as 1 KiB per instruction (intfft_fast1024.hip does the same)
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4i y = {re[q], im[q], re[q + 8], im[q + 8]};
                    *reinterpret_cast<v4i *>(lds + 2 * (lane_off + (q & 1) * out_weight<L>(0) + ((q >> 1) & 1) * out_weight<L>(1) + (q >> 2) * out_weight<L>(2))) = y;
                }
                wave_lds_fence();
                v4i *dst4 = reinterpret_cast<v4i *>(out + f * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 64 * i + lane; // 16-byte piece = two samples: frame (2 e) >> 6 within the chunk
                    if (f * FP + (size_t)(e >> (L - 1)) >= nframes_user) continue;
                    __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 4 * e), dst4 + e);
                }
                wave_lds_fence();
            } else if (f * FP + (size_t)lane_frame < nframes_user) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4i y = {re[q], im[q], re[q + 8], im[q + 8]};
                    __builtin_nontemporal_store(y, reinterpret_cast<v4i *>(dst + (q & 1) * out_weight<L>(0) +
                                                                           ((q >> 1) & 1) * out_weight<L>(1) +
                                                                           (q >> 2) * out_weight<L>(2)));
                }
            }
        } else {
            int2 *dst = out + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i y = {re[r], im[r]};
                __builtin_nontemporal_store(y, reinterpret_cast<v2i *>(dst + 64 * rr));
            }
        }
    }
}

bool fast1024u_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly,
                         int in_order, int out_order)
{
    if (!(log2n >= 6 && log2n <= 10 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 1 && direction == 0 && use_fly == 1))
        return false;
    if (in_order == 0 && out_order == 0) return true;
    // int_fftNk's own beat orders (HALVES in, BITREV out) and the mixed forms, N >= 128
    return log2n >= 7 && (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1);
}

const char *fast1024u_kernel_name() { return "k_fft1024_u32"; }

template <int L, bool FAST_OK, bool NAT>
static hipError_t launchu(const u32 *in, int2 *out, const int2 *tw, const UConsts &c, size_t nframes, int sh,
                          hipStream_t stream, int native)
{
    const size_t cap = resident_blocks(kptr(k_fft1024_u32<L, FAST_OK, NAT>), 256, 2);
    const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L);
    const size_t need = (chunks + 3) / 4;
    hipLaunchKernelGGL((k_fft1024_u32<L, FAST_OK, NAT>), dim3((unsigned)(need < cap ? need : cap)), dim3(256), 0, stream, in, out, tw,
                       c, nframes, sh, native);
    return hipGetLastError();
}

template <int L>
static hipError_t launchu_l(bool fast, const u32 *in, int2 *out, const int2 *tw, const UConsts &c, size_t nframes, int sh,
                            hipStream_t stream, int native)
{
    if constexpr (L >= 7) {
        if (native) return fast ? launchu<L, true, true>(in, out, tw, c, nframes, sh, stream, native) : launchu<L, false, true>(in, out, tw, c, nframes, sh, stream, native);
    }
    return fast ? launchu<L, true, false>(in, out, tw, c, nframes, sh, stream, 0) : launchu<L, false, false>(in, out, tw, c, nframes, sh, stream, 0);
}

hipError_t launch_fast1024u(int log2n, int twd, const void *in, void *out, const int2 *tw_all, const int2 *h_tw, size_t nframes,
                            hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) {
        c.wr3[k] = h_tw[7 + k].x;
        c.wi3[k] = h_tw[7 + k].y;
    }
    for (int k = 0; k < 4; ++k) {
        c.wr2[k] = h_tw[3 + k].x;
        c.wi2[k] = h_tw[3 + k].y;
    }
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const u32 *pin = static_cast<const u32 *>(in);
    int2 *pout = static_cast<int2 *>(out);
    switch (log2n) {
    case 6: return launchu_l<6>(allow_fast, pin, pout, tw_all, c, nframes, twd - 1, stream, native);
    case 7: return launchu_l<7>(allow_fast, pin, pout, tw_all, c, nframes, twd - 1, stream, native);
    case 8: return launchu_l<8>(allow_fast, pin, pout, tw_all, c, nframes, twd - 1, stream, native);
    case 9: return launchu_l<9>(allow_fast, pin, pout, tw_all, c, nframes, twd - 1, stream, native);
    default: return launchu_l<10>(allow_fast, pin, pout, tw_all, c, nframes, twd - 1, stream, native);
    }
}

} // namespace intfft
