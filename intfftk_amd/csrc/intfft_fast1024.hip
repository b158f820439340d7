// intfft_fast1024.hip -- packed-int16 wave kernel for the headline configuration:
// int_fftNk with NFFT = 10 (N = 1024), DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled (FORMAT = 0),
// natural-order input, NATURAL or BITREV output  (src/vhdl/main/int_fft_single_path.vhd:157-268).
//
// One wave64 owns one frame: 16 VGPRs of packed (re | im << 16) int16 per lane, persistent loop
// over frames.  The ten radix-2 DIF stages (src/vhdl/fft/int_dif2_fly.vhd:144-373) are evaluated
// literally -- per-stage truncation forbids any algebraic stage fusion -- but the DATAFLOW is
// regrouped so that every butterfly is lane-local:
//
//   index bit:        9 8 7 6 | 5      | 4      | 3 2 1 0
//   phase 1 (regs)    j3..j0  | lane5  | lane4  | lane3..0     stages 9,8,7,6 in registers
//   v_permlane32_swap lane5   | j3     |                       stage 5
//   v_permlane16_swap         |        | j2                    stage 4
//   LDS transpose (5 KiB/wave, conflict-free b32 writes, b128 reads): regs = bits 3..0
//   phase 3 (regs)                                 r3..r0      stages 3,2 (wave-uniform twiddles
//                                                              in SGPRs), 1, 0 (multiplier-free)
//
// The cross-commutators (src/vhdl/delay/int_delay_line.vhd:60-104) are exactly this regrouping;
// the final bit-reversal (src/vhdl/buffers/int_bitrev_order.vhd:82-104) is folded into the LDS
// transpose so that every global store instruction writes 256 contiguous bytes.
//
// Arithmetic per general butterfly (SURVEY.md section 9.2, 9.4 "sngl" regime, w = 16):
//   A1 = A >> 1, B1 = B >> 1            v_pk_ashrrev_i16 x2   (LSB dropped BEFORE the add)
//   S = A1 + B1, D = A1 - B1            v_pk_add_u16, v_pk_sub_i16
//   re = D.re*wr - D.im*wi              v_dot2_i32_i16 with W packed as (wr, -wi)   (exact in int32)
//   im = D.re*wi + D.im*wr              v_dot2_i32_i16 with W packed as (wi,  wr)
//   Y  = { im[t+14:t-1], re[t+14:t-1] } 2 shifts + v_perm_b32   (floor, wrap to 16 bits)
#include "intfft_internal.hpp"

namespace intfft {

using u32 = uint32_t;
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(u32 x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ u32 as_u32(v2s x) { return __builtin_bit_cast(u32, x); }

// wave-uniform twiddles of the in-register stages 3 and 2, both packings (kernel argument -> SGPRs)
struct Fast1024Consts {
    u32 wa3[8], wb3[8]; // STAGE 3: table index r & 7
    u32 wa2[4], wb2[4]; // STAGE 2: table index r & 3
};

template <bool ROUND> __device__ __forceinline__ void sumdiff(u32 a, u32 b, u32 &s, u32 &d)
{
    const v2s A = as_v2s(a), B = as_v2s(b);
    if (!ROUND) { // int_dif2_fly.vhd:144-164
        const v2s A1 = A >> (short)1, B1 = B >> (short)1;
        s = as_u32(A1 + B1);
        d = as_u32(A1 - B1);
    } else { // :167-219  rhu2(A+B) = (A|B) - ((A^B)>>1);  rhu2(A-B) = (A>>1) - (B>>1) + (A & ~B & 1)
        s = as_u32((A | B) - ((A ^ B) >> (short)1));
        const v2s one = {1, 1};
        d = as_u32((A >> (short)1) - (B >> (short)1) + ((A & ~B) & one));
    }
}

// cmult_{16,t}(D, W) in the single-DSP regime (int_cmult_dsp48.vhd:184-225)
__device__ __forceinline__ u32 cmul(u32 d, u32 wa, u32 wb, int sh_r, int sh_l)
{
    const int re = __builtin_amdgcn_sdot2(as_v2s(d), as_v2s(wa), 0, false);
    const int im = __builtin_amdgcn_sdot2(as_v2s(d), as_v2s(wb), 0, false);
    return __builtin_amdgcn_perm((u32)im << sh_l, (u32)re >> sh_r, 0x07060100u);
}

template <bool ROUND>
__device__ __forceinline__ void bfly(u32 &a, u32 &b, u32 wa, u32 wb, int sh_r, int sh_l)
{
    u32 s, d;
    sumdiff<ROUND>(a, b, s, d);
    a = s;
    b = cmul(d, wa, wb, sh_r, sh_l);
}

// STAGE 0 and even positions of STAGE 1: Y = D (int_dif2_fly.vhd:245-255, :293-296)
template <bool ROUND> __device__ __forceinline__ void bfly_triv(u32 &a, u32 &b)
{
    u32 s, d;
    sumdiff<ROUND>(a, b, s, d);
    a = s;
    b = d;
}

// odd positions of STAGE 1: Y.re = D.im, Y.im = D.re >= 0 ? -D.re : ~D.re (int_dif2_fly.vhd:297-304)
template <bool ROUND> __device__ __forceinline__ void bfly_mj(u32 &a, u32 &b)
{
    u32 s, d;
    sumdiff<ROUND>(a, b, s, d);
    a = s;
    const u32 rot = __builtin_amdgcn_alignbit(d, d, 16); // lo = D.im, hi = D.re
    const u32 nx = rot ^ 0xFFFF0000u;                     // hi = ~D.re
    b = nx + ((nx >> 31) << 16);                          // + 1 in the high half iff D.re >= 0
}

__device__ __forceinline__ u32 pack_wa(int2 w) { return ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16); }
__device__ __forceinline__ u32 pack_wb(int2 w) { return ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16); }

constexpr int ROW_DW = 20; // LDS row stride in dwords: 16 data + 4 pad (16-B aligned, conflict-free)

template <bool ROUND, bool OUT_BITREV>
__global__ __launch_bounds__(256) void k_fft1024_i16(const u32 *in, u32 *out, const int2 *__restrict__ tw,
                                                     const Fast1024Consts c, size_t nframes, int sh_r,
                                                     int sh_l)
{
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * 64 * ROW_DW];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    u32 *lds = lds_all + wv * 64 * ROW_DW;

    // ---- per-lane twiddles of the lane-dependent stages (frame invariant) --------------------
    // stage s table starts at tw + 2^s - 1; index = position mod 2^s (rom_twiddle_int.vhd:187-202)
    u32 wa9[8], wb9[8], wa8[4], wb8[4], wa7[2], wb7[2], wa6, wb6, wa5, wb5, wa4, wb4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int2 w = tw[511 + 64 * j + lane];
        wa9[j] = pack_wa(w);
        wb9[j] = pack_wb(w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int2 w = tw[255 + 64 * j + lane];
        wa8[j] = pack_wa(w);
        wb8[j] = pack_wb(w);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int2 w = tw[127 + 64 * j + lane];
        wa7[j] = pack_wa(w);
        wb7[j] = pack_wb(w);
    }
    {
        int2 w = tw[63 + lane];
        wa6 = pack_wa(w);
        wb6 = pack_wb(w);
        w = tw[31 + (lane & 31)];
        wa5 = pack_wa(w);
        wb5 = pack_wb(w);
        w = tw[15 + (lane & 15)];
        wa4 = pack_wa(w);
        wb4 = pack_wb(w);
    }

    // ---- LDS transpose addressing -------------------------------------------------------------
    // after the two lane swaps: lane5 = n9, lane4 = n8, lane3..0 = n3..0; reg j3 = n5, j2 = n4,
    // j1 = n7, j0 = n6.  Destination row = new lane:
    //   NATURAL: row bit i = n(9-i)  (so that X index = rev4(r) * 64 + row, contiguous in row)
    //   BITREV : row = n9..n4        (memory index = n = row * 16 + r)
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    const int wr_lane = OUT_BITREV ? ROW_DW * (32 * t5 + 16 * t4) + (lane & 15)
                                   : ROW_DW * (t5 + 2 * t4) + (lane & 15);
    u32 *wr_base = lds + wr_lane;
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROW_DW * lane);

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv;
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        const u32 *src = in + f * 1024 + lane;
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[64 * j]; // lane = n5..0, j = n9..6

        // ---- phase 1: stages 9, 8, 7, 6 ----
#pragma unroll
        for (int j = 0; j < 8; ++j) bfly<ROUND>(v[j], v[j + 8], wa9[j], wb9[j], sh_r, sh_l);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) bfly<ROUND>(v[g + j], v[g + j + 4], wa8[j], wb8[j], sh_r, sh_l);
#pragma unroll
        for (int g = 0; g < 16; g += 4)
#pragma unroll
            for (int j = 0; j < 2; ++j) bfly<ROUND>(v[g + j], v[g + j + 2], wa7[j], wb7[j], sh_r, sh_l);
#pragma unroll
        for (int g = 0; g < 16; g += 2) bfly<ROUND>(v[g], v[g + 1], wa6, wb6, sh_r, sh_l);

        // ---- lane bit 5 <-> reg bit 3, stage 5 ----
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const auto r = __builtin_amdgcn_permlane32_swap(v[j], v[j + 8], false, false);
            v[j] = r[0];
            v[j + 8] = r[1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) bfly<ROUND>(v[j], v[j + 8], wa5, wb5, sh_r, sh_l);

        // ---- lane bit 4 <-> reg bit 2, stage 4 ----
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const auto r = __builtin_amdgcn_permlane16_swap(v[g + j], v[g + j + 4], false, false);
                v[g + j] = r[0];
                v[g + j + 4] = r[1];
            }
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) bfly<ROUND>(v[g + j], v[g + j + 4], wa4, wb4, sh_r, sh_l);

        // ---- LDS transpose: regs become n3..0 ----
        asm volatile("" ::: "memory"); // keep the previous frame's reads ahead of these writes
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;
            const int row_j = OUT_BITREV ? (8 * j1 + 4 * j0 + 2 * j3 + j2) : (4 * j1 + 8 * j0 + 16 * j3 + 32 * j2);
            wr_base[ROW_DW * row_j] = v[j];
        }
        asm volatile("" ::: "memory"); // LDS ops of one wave execute in order: no barrier needed
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd_base[q];
            v[4 * q + 0] = x.x;
            v[4 * q + 1] = x.y;
            v[4 * q + 2] = x.z;
            v[4 * q + 3] = x.w;
        }
        asm volatile("" ::: "memory");

        // ---- phase 3: stages 3, 2 (uniform twiddles), 1, 0 ----
#pragma unroll
        for (int r = 0; r < 8; ++r) bfly<ROUND>(v[r], v[r + 8], c.wa3[r], c.wb3[r], sh_r, sh_l);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) bfly<ROUND>(v[g + r], v[g + r + 4], c.wa2[r], c.wb2[r], sh_r, sh_l);
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            bfly_triv<ROUND>(v[g], v[g + 2]);
            bfly_mj<ROUND>(v[g + 1], v[g + 3]);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) bfly_triv<ROUND>(v[g], v[g + 1]);

        // ---- store ----
        if (OUT_BITREV) {
            uint4 *dst = reinterpret_cast<uint4 *>(out + f * 1024 + lane * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        } else {
            u32 *dst = out + f * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); // rev4
                dst[64 * rr] = v[r];
            }
        }
    }
}

bool fast1024_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction,
                        int use_fly, int in_order, int out_order)
{
    (void)rndmode;
    return log2n == 10 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 &&
           direction == 0 && use_fly == 1 && in_order == 0 && (out_order == 0 || out_order == 1);
}

const char *fast1024_kernel_name() { return "k_fft1024_i16"; }

template <bool ROUND, bool OUT_BITREV>
static hipError_t launch_t(const u32 *in, u32 *out, const int2 *tw, const Fast1024Consts &c, size_t nframes,
                           int sh_r, int sh_l, hipStream_t stream)
{
    // persistent waves: 8 blocks of 4 waves per CU at most (LDS: 8 x 20 KiB = 160 KiB)
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t need = (nframes + 3) / 4;
    const size_t cap = (size_t)cus * 8;
    const unsigned blocks = (unsigned)(need < cap ? need : cap);
    hipLaunchKernelGGL((k_fft1024_i16<ROUND, OUT_BITREV>), dim3(blocks), dim3(256), 0, stream, in, out, tw, c,
                       nframes, sh_r, sh_l);
    return hipGetLastError();
}

hipError_t launch_fast1024(const Fast1024Args &a, const void *in, void *out, const int2 *tw_all,
                           const int2 *h_tw, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Fast1024Consts c;
    for (int k = 0; k < 8; ++k) { // stage 3 table at offset 2^3 - 1
        const int2 w = h_tw[7 + k];
        c.wa3[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb3[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    for (int k = 0; k < 4; ++k) { // stage 2 table at offset 3
        const int2 w = h_tw[3 + k];
        c.wa2[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb2[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    const int sh_r = a.twd - 1, sh_l = 17 - a.twd;
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    if (a.rnd == RND_ROUND)
        return a.out_bitrev ? launch_t<true, true>(pin, pout, tw_all, c, nframes, sh_r, sh_l, stream)
                            : launch_t<true, false>(pin, pout, tw_all, c, nframes, sh_r, sh_l, stream);
    return a.out_bitrev ? launch_t<false, true>(pin, pout, tw_all, c, nframes, sh_r, sh_l, stream)
                        : launch_t<false, false>(pin, pout, tw_all, c, nframes, sh_r, sh_l, stream);
}

} // namespace intfft
