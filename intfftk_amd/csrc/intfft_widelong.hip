// intfft_widelong.hip -- int_fftNk with FORMAT = 1 (full bit growth) at N = 2^17 .. 2^20 (round 5): 16-bit ADC data through a long unscaled core,
// 33 .. 36-bit results (and DATA_WIDTH 17 .. 20+ in int32 containers while DATA_WIDTH + NFFT <= 40), natural order or the core's own beat orders (HALVES in: one
// access per block pair in k_wide_pre; BITREV out: the NAT instantiations of k_wide16_p2).  These lengths ran the
// generic k_pass<int64> passes (three or four passes on 16-byte words: ~50 Gsample/s); the 24-bit class of intfft_wide16.hip stops at N = 2^16.
//
// N = 2^LX = B blocks of 2^16 points, B = 2^XS, n = 65536 b + 256 r + c (int_fftNk.vhd:184-342: the DIF stages run STAGE LX-1 .. 0, stage ii has
// DATA_WIDTH + ii bit inputs):
//   pass 0  k_wide_pre<XS>            STAGE LX-1 .. 16 (over b; widths <= DATA_WIDTH + 4): int32 registers, user array -> scratch half A (int32 pairs, in place)
//   pass 1  k_wide16_p1<16, ., XS>    STAGE 15 .. 8 of every block (over r; widths <= 32): the 24-bit class's first pass, scratch A -> scratch B
//   pass 2  k_wide16_p2<16, ., ., XS> STAGE 7 .. 0 (over c; 64-bit words) on units that take their 16 rows ACROSS the blocks, so that the natural-order store
//                                     X[brev_LX(n)] still writes 256-byte runs: scratch B -> user array.  16-bit data stays within 32 bits up to STAGE 4
//                                     (DATA_WIDTH + LX - 4 <= 32): its round 1 runs on the int32 butterflies (R32 instantiations)
// Pass traffic: 4 + 8, 8 + 8, 8 + 16 = 52 B/sample (int16 containers in) against 20 algorithmic.
// The twiddle of STAGE s at position n is table entry 2^s - 1 + (n mod 2^s) (rom_twiddle_int.vhd / row_twiddle_tay.vhd through k_twiddle_stage): in pass 0 it
// depends on (b below the stage's bit, r, c) -- frame invariant, held in VGPRs over the workgroup's frame loop.
#define INTFFT_NT_LOADS 1
#define INTFFT_WIDE16_TEMPLATES_ONLY 1
#include "intfft_wide16.hip"

namespace intfft {

// the butterfly of pass 0: wfly32 with the general form of the result slice -- bits [sh, sh + wo) of the 64-bit sum for any sh <= 31, wo <= 32 (wfly32's
// v_alignbit + one shift needs sh + wo >= 32, which 16-bit data under 16-bit twiddles only just meets and narrower data does not); one shift more per
// component, in a pass that waits for memory
__device__ __forceinline__ void wfly32g(int &are, int &aim, int &bre, int &bim, int wr, int wi, const WideStage &s)
{
    asm volatile("" : "+v"(wr), "+v"(wi));
    const int dre = are - bre, dim = aim - bim; // unscaled: exact, one bit of growth (int_dif2_fly.vhd:222-240)
    are += bre;
    aim += bim;
    const u64 m2r = (u64)((i64)dre * wr), m1r = (u64)((i64)dim * wi);
    const u64 m2i = (u64)((i64)dre * wi), m1i = (u64)((i64)dim * wr);
    const u64 k = 0xFFFFFFFF00000000ull | s.keep; // (M >> a) << a == M & ~(2^a - 1): int_cmult_dsp48.vhd:182-434
    const u64 xr = (m2r & k) - (m1r & k), xi = (m2i & k) + (m1i & k);
    bre = (int)(__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.sh) << s.s3) >> s.s3;
    bim = (int)(__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.sh) << s.s3) >> s.s3;
}

// ---- pass 0: STAGE LX-1 .. 16 ---------------------------------------------------------------------------------------------------
// thread = P positions p = 256 P tile + 256 i + tid (i < P, P B = 16: sixteen samples per thread at every length), registers [i][b]; every global
// instruction of a wave covers 256 (int16 containers) / 512 contiguous bytes.  Grid = (frame groups) x (256 / P tiles); a workgroup keeps its tile.
template <int XS, bool IN16>
__global__ __launch_bounds__(256) void k_wide_pre(const void *in, int2 *scr, const int2 *__restrict__ twt, const WideArgs a, size_t nframes)
{
    static_assert(XS >= 1 && XS <= 4, "N = 2^17 .. 2^20");
    constexpr int LX = 16 + XS, B = 1 << XS, P = 16 >> XS, TILES = 256 / P;
    const int tid = threadIdx.x;
    const unsigned tile = blockIdx.x % TILES;
    const unsigned p0 = 256u * P * tile + (unsigned)tid;
    // twiddles: stage ii (STAGE LX-1-ii), pair (b, b + H), H = B >> (ii + 1): index (b mod H) 65536 + p
    int wr[P][B - 1 > 0 ? B - 1 : 1], wi[P][B - 1 > 0 ? B - 1 : 1]; // [i][H - 1 + j]: the stages' sets back to back, top stage last
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int H = 1; H < B; H <<= 1) {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int2 w = twt[(size_t)65536 * H - 1 + (size_t)65536 * j + p0 + 256u * i];
                wr[i][H - 1 + j] = w.x, wi[i][H - 1 + j] = w.y;
            }
        }
    }
    const size_t fstep = gridDim.x / TILES;
    for (size_t f = blockIdx.x / TILES; f < nframes; f += fstep) {
        int re[16], im[16];
        unsigned toff = p0;
        asm volatile("" : "+v"(toff)); // opaque per iteration (see k_wide16_p1)
        if (a.native & 1) { // HALVES order in (int_fftNk.vhd:15-21): beat (x[i], x[i + N/2]) = the blocks (b, b + B/2) at one position: adjacent samples, ONE access
            if constexpr (IN16) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                const v2u *src = static_cast<const v2u *>(in) + (f << (LX - 1));
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int b = 0; b < B / 2; ++b) {
                        const v2u x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                        re[i * B + b] = __builtin_amdgcn_sbfe((int)x.x, 0, a.dw), im[i * B + b] = __builtin_amdgcn_sbfe((int)x.x, 16, a.dw);
                        re[i * B + b + B / 2] = __builtin_amdgcn_sbfe((int)x.y, 0, a.dw), im[i * B + b + B / 2] = __builtin_amdgcn_sbfe((int)x.y, 16, a.dw);
                    }
            } else {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i *src = static_cast<const v4i *>(in) + (f << (LX - 1));
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int b = 0; b < B / 2; ++b) {
                        const v4i x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                        re[i * B + b] = __builtin_amdgcn_sbfe(x.x, 0, a.dw), im[i * B + b] = __builtin_amdgcn_sbfe(x.y, 0, a.dw);
                        re[i * B + b + B / 2] = __builtin_amdgcn_sbfe(x.z, 0, a.dw), im[i * B + b + B / 2] = __builtin_amdgcn_sbfe(x.w, 0, a.dw);
                    }
            }
        } else if constexpr (IN16) {
            const u32 *src = static_cast<const u32 *>(in) + (f << LX);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const u32 x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                    re[i * B + b] = __builtin_amdgcn_sbfe((int)x, 0, a.dw); // conv_std_logic_vector(.., DATA_WIDTH): wrap on load
                    im[i * B + b] = __builtin_amdgcn_sbfe((int)x, 16, a.dw);
                }
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + (f << LX);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const v2i x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                    re[i * B + b] = __builtin_amdgcn_sbfe(x.x, 0, a.dw);
                    im[i * B + b] = __builtin_amdgcn_sbfe(x.y, 0, a.dw);
                }
        }
#pragma unroll
        for (int ii = 0; ii < XS; ++ii) {
            const int H = B >> (ii + 1);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int g = 0; g < B; g += 2 * H)
#pragma unroll
                    for (int j = 0; j < H; ++j)
                        wfly32g(re[i * B + g + j], im[i * B + g + j], re[i * B + g + j + H], im[i * B + g + j + H], wr[i][H - 1 + j], wi[i][H - 1 + j], a.st[ii]);
        }
        typedef int v2i __attribute__((ext_vector_type(2)));
        v2i *dst = reinterpret_cast<v2i *>(scr) + (f << LX);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const v2i y = {re[i * B + b], im[i * B + b]};
                *at32(dst + 65536 * b + 256 * i, toff) = y;
            }
    }
}

template <int XS>
static hipError_t launch_long(const WideArgs &a, const W2Consts &k, bool in16, const void *in, void *out, void *scratch, const int2 *tw_all, size_t nframes,
                              hipStream_t stream)
{
    constexpr int LX = 16 + XS, TILES = 256 / (16 >> XS);
    int2 *const scr_a = static_cast<int2 *>(scratch);
    int2 *const scr_b = scr_a + (nframes << LX); // the plan's scratch frames: an int32-pair part A, then part B (int32 pairs; 64-bit pairs behind k_wide64_p1)
    {
        const void *kern = in16 ? kptr(k_wide_pre<XS, true>) : kptr(k_wide_pre<XS, false>);
        size_t g = resident_blocks(kern, 256, 4) / TILES;
        if (g < 1) g = 1;
        if (g > nframes) g = nframes;
        if (in16)
            hipLaunchKernelGGL((k_wide_pre<XS, true>), dim3((unsigned)(g * TILES)), dim3(256), 0, stream, in, scr_a, tw_all, a, nframes);
        else
            hipLaunchKernelGGL((k_wide_pre<XS, false>), dim3((unsigned)(g * TILES)), dim3(256), 0, stream, in, scr_a, tw_all, a, nframes);
    }
    const size_t nblocks = nframes << XS;
    if (a.w64) { // DATA_WIDTH + NFFT - 8 > 32: STAGE 15 .. 8 on 64-bit words too (the class of k_wide64_p1), 16-byte samples in part B
        WideArgs a1 = a;
        a1.dw = a.dw + XS, a1.native = 0;
        const size_t units = nblocks * 16;
        size_t g = resident_blocks(kptr(k_wide64_p1<16, false, XS>), 256, 2) & ~(size_t)15;
        if (g < 16) g = 16;
        if (g > units) g = units;
        hipLaunchKernelGGL((k_wide64_p1<16, false, XS>), dim3((unsigned)g), dim3(256), 0, stream, scr_a, reinterpret_cast<i64 *>(scr_b), tw_all, a1, nblocks);
        const size_t units2 = nframes << (4 + XS);
        size_t g2 = resident_blocks(kptr(k_wide16_p2<16, true, false, XS>), 256, 2);
        if (g2 > units2) g2 = units2;
        if (a.native & 2) // BITREV order out: the NAT instantiation (16 x 16 exchange through the planes, rows of the unit across the blocks)
            hipLaunchKernelGGL((k_wide16_p2<16, true, true, XS>), dim3((unsigned)g2), dim3(256), 0, stream, scr_b, static_cast<i64 *>(out), tw_all, a, k, nframes);
        else
            hipLaunchKernelGGL((k_wide16_p2<16, true, false, XS>), dim3((unsigned)g2), dim3(256), 0, stream, scr_b, static_cast<i64 *>(out), tw_all, a, k, nframes);
        return hipGetLastError();
    }
    {
        WideArgs a1 = a;
        a1.dw = a.dw + XS, a1.native = 0; // what the pre-pass wrote: values of DATA_WIDTH + XS bits (the wrap on load is then the identity) at plain core positions
        const size_t units = nblocks * 16;
        size_t g = resident_blocks(kptr(k_wide16_p1<16, false, XS>), 256, 2) & ~(size_t)15;
        if (g < 16) g = 16;
        if (g > units) g = units;
        hipLaunchKernelGGL((k_wide16_p1<16, false, XS>), dim3((unsigned)g), dim3(256), 0, stream, scr_a, scr_b, tw_all, a1, nblocks);
    }
    {
        const size_t units = nframes << (4 + XS);
        size_t g = resident_blocks(a.r32 ? kptr(k_wide16_p2<16, false, false, XS, true>) : kptr(k_wide16_p2<16, false, false, XS>), 256, 2);
        if (g > units) g = units;
        if (a.native & 2) {
            if (a.r32)
                hipLaunchKernelGGL((k_wide16_p2<16, false, true, XS, true>), dim3((unsigned)g), dim3(256), 0, stream, scr_b, static_cast<i64 *>(out), tw_all, a, k, nframes);
            else
                hipLaunchKernelGGL((k_wide16_p2<16, false, true, XS>), dim3((unsigned)g), dim3(256), 0, stream, scr_b, static_cast<i64 *>(out), tw_all, a, k, nframes);
        } else if (a.r32)
            hipLaunchKernelGGL((k_wide16_p2<16, false, false, XS, true>), dim3((unsigned)g), dim3(256), 0, stream, scr_b, static_cast<i64 *>(out), tw_all, a, k, nframes);
        else
            hipLaunchKernelGGL((k_wide16_p2<16, false, false, XS>), dim3((unsigned)g), dim3(256), 0, stream, scr_b, static_cast<i64 *>(out), tw_all, a, k, nframes);
    }
    return hipGetLastError();
}

// ---- the inverse of class 1 (int_ifftNk.vhd:183-341, DIT STAGE 0 .. LX-1): k_wide16_q1<16, ., XS> (bit-reversed gather at L = LX + STAGE 0 .. 7 on int32, rows of a
// unit across the blocks), k_wide16_q2<16> on the blocks (STAGE 8 .. 15 on 64-bit words; its natural-order store goes to scratch part C), then STAGE 16 .. LX-1 here on
// 64-bit words: scratch C -> user array, natural order.  Thread / tile shape of k_wide_pre.  Pass traffic 4 + 8, 8 + 16, 16 + 16 = 68 B/sample (int16 containers in).
template <int XS>
__global__ __launch_bounds__(256) void k_wide_post(const i64 *scr, i64 *out, const int2 *__restrict__ twt, const WideArgs a, size_t nframes)
{
    static_assert(XS >= 1 && XS <= 4, "N = 2^17 .. 2^20");
    constexpr int LX = 16 + XS, B = 1 << XS, P = 16 >> XS, TILES = 256 / P;
    const int tid = threadIdx.x;
    const unsigned tile = blockIdx.x % TILES;
    const unsigned p0 = 256u * P * tile + (unsigned)tid;
    int wr[P][B - 1], wi[P][B - 1]; // [i][H - 1 + j]: STAGE 16 + log2 H pairs (b, b + H), index (b mod H) 65536 + p
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int H = 1; H < B; H <<= 1)
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int2 w = twt[(size_t)65536 * H - 1 + (size_t)65536 * j + p0 + 256u * i];
                wr[i][H - 1 + j] = w.x, wi[i][H - 1 + j] = w.y;
            }
    typedef i64 v2l __attribute__((ext_vector_type(2)));
    const size_t fstep = gridDim.x / TILES;
    for (size_t f = blockIdx.x / TILES; f < nframes; f += fstep) {
        i64 re[16], im[16];
        unsigned toff = p0;
        asm volatile("" : "+v"(toff));
        const v2l *src = reinterpret_cast<const v2l *>(scr) + (f << LX);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const v2l x = INTFFT_LD(at32(src + 65536 * b + 256 * i, toff));
                re[i * B + b] = x.x, im[i * B + b] = x.y;
            }
#pragma unroll
        for (int ii = 0; ii < XS; ++ii) {
            const int H = 1 << ii;
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int g = 0; g < B; g += 2 * H)
#pragma unroll
                    for (int j = 0; j < H; ++j)
                        wdit64<false>(re[i * B + g + j], im[i * B + g + j], re[i * B + g + j + H], im[i * B + g + j + H], wr[i][H - 1 + j], wi[i][H - 1 + j], a.st[16 + ii]);
        }
        if (a.native & 1) { // HALVES order out (int_ifftNk.vhd:15-21): the blocks (b, b + B/2) of one position leave as one 32-byte access
            typedef i64 v4l __attribute__((ext_vector_type(4)));
            v4l *dst4 = reinterpret_cast<v4l *>(out) + (f << (LX - 1));
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int b = 0; b < B / 2; ++b) {
                    const v4l y = {re[i * B + b], im[i * B + b], re[i * B + b + B / 2], im[i * B + b + B / 2]};
                    __builtin_nontemporal_store(y, at32(dst4 + 65536 * b + 256 * i, toff));
                }
            continue;
        }
        v2l *dst = reinterpret_cast<v2l *>(out) + (f << LX);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const v2l y = {re[i * B + b], im[i * B + b]};
                __builtin_nontemporal_store(y, at32(dst + 65536 * b + 256 * i, toff));
            }
    }
}

template <int XS>
static hipError_t launch_long_inv(const WideArgs &a, const W2Consts &k, bool in16, const void *in, void *out, void *scratch, const int2 *tw_all, size_t nframes,
                                  hipStream_t stream)
{
    constexpr int LX = 16 + XS, TILES = 256 / (16 >> XS);
    int2 *const scr_a = static_cast<int2 *>(scratch);                    // k_wide16_q1's output: int32 pairs in the blocks' unit layouts (class 2: 64-bit pairs)
    i64 *const scr_c = reinterpret_cast<i64 *>(scr_a + ((nframes << LX) << (a.w64 ? 1 : 0))); // k_wide16_q2's output: 64-bit pairs at the core positions
    const size_t nblocks = nframes << XS;
    if (a.w64) { // class 2: DATA_WIDTH + 8 > 32 or twiddles below 16 bits -- STAGE 0 .. 7 on 64-bit words too (k_wide64_q1), 16-byte samples in part A
        const size_t units = nframes << (4 + XS);
        size_t g = resident_blocks(kptr(k_wide64_q1<16, false, XS>), 256, 2);
        if (g > units) g = units;
        if (a.native & 2) // BITREV order in: the NAT instantiation (rows of a unit across the blocks)
            hipLaunchKernelGGL((k_wide64_q1<16, true, XS>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const int2 *>(in), reinterpret_cast<i64 *>(scr_a), tw_all, a, k, nframes);
        else
            hipLaunchKernelGGL((k_wide64_q1<16, false, XS>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const int2 *>(in), reinterpret_cast<i64 *>(scr_a), tw_all, a, k, nframes);
        WideArgs a1 = a;
        a1.native = 0;
        const size_t units2 = nblocks * 16;
        size_t g2 = resident_blocks(kptr(k_wide16_q2<16, true, false>), 256, 2) & ~(size_t)15;
        if (g2 < 16) g2 = 16;
        if (g2 > units2) g2 = units2;
        hipLaunchKernelGGL((k_wide16_q2<16, true, false>), dim3((unsigned)g2), dim3(256), 0, stream, scr_a, scr_c, tw_all, a1, nblocks);
    } else {
    {
        const size_t units = nframes << (4 + XS);
        size_t g = resident_blocks(in16 ? kptr(k_wide16_q1<16, false, XS, true>) : kptr(k_wide16_q1<16, false, XS, false>), 256, 2);
        if (g > units) g = units;
        if (a.native & 2) { // BITREV order in
            if (in16)
                hipLaunchKernelGGL((k_wide16_q1<16, true, XS, true>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const int2 *>(in), scr_a, tw_all, a, k, nframes);
            else
                hipLaunchKernelGGL((k_wide16_q1<16, true, XS, false>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const int2 *>(in), scr_a, tw_all, a, k, nframes);
        } else if (in16)
            hipLaunchKernelGGL((k_wide16_q1<16, false, XS, true>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const int2 *>(in), scr_a, tw_all, a, k, nframes);
        else
            hipLaunchKernelGGL((k_wide16_q1<16, false, XS, false>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const int2 *>(in), scr_a, tw_all, a, k, nframes);
    }
    {
        WideArgs a1 = a;
        a1.native = 0;
        const size_t units = nblocks * 16;
        size_t g = resident_blocks(kptr(k_wide16_q2<16, false, false>), 256, 2) & ~(size_t)15;
        if (g < 16) g = 16;
        if (g > units) g = units;
        hipLaunchKernelGGL((k_wide16_q2<16, false, false>), dim3((unsigned)g), dim3(256), 0, stream, scr_a, scr_c, tw_all, a1, nblocks);
    }
    }
    {
        size_t g = resident_blocks(kptr(k_wide_post<XS>), 256, 3) / TILES;
        if (g < 1) g = 1;
        if (g > nframes) g = nframes;
        hipLaunchKernelGGL((k_wide_post<XS>), dim3((unsigned)(g * TILES)), dim3(256), 0, stream, scr_c, static_cast<i64 *>(out), tw_all, a, nframes);
    }
    return hipGetLastError();
}

int widelong_class(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order)
{
    if (direction == 1) { // int_ifftNk (round 5)
        if (!(log2n >= 17 && log2n <= 20 && format == 1 && use_fly == 1 && (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2) && data_width >= 9 &&
              data_width + log2n > 32))
            return 0; // NATURAL | BITREV in, NATURAL | HALVES out
        // class 1: STAGE 0 .. 7 within int32, results of 33 .. 40 bits; class 2: every stage on 64-bit words (int32 containers in), results up to 48 bits
        if (data_width + 8 <= 32 && data_width + log2n <= 40 && twdl_width >= 16 && twdl_width <= 24) return 1;
        return data_width >= 17 && data_width <= 32 && data_width + log2n <= 48 && twdl_width >= 8 && twdl_width <= 24 ? 2 : 0;
    }
    if (!(log2n >= 17 && log2n <= 20 && format == 1 && direction == 0 && use_fly == 1 && (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1) &&
          data_width >= 9))
        return 0; // int_fftNk: NATURAL | HALVES in, NATURAL | BITREV out
    const int xs = log2n - 16;
    if (data_width + log2n <= 32) return 0; // results in int32 containers: not this class
    // class 1: pass 1 within int32 (DATA_WIDTH + NFFT - 8 <= 32), results of 33 .. 40 bits (the 24-bit class of k_wide16_p1 / p2)
    if (data_width + log2n - 8 <= 32 && data_width + log2n <= 40 && twdl_width >= 16 && twdl_width <= 24) return 1;
    // class 2: pass 0 within int32 (DATA_WIDTH + XS <= 32), passes 1 and 2 on 64-bit words (the class of k_wide64_p1), results up to 48 bits
    if (data_width >= 17 && data_width + xs <= 32 && data_width + log2n <= 48 && twdl_width >= 8 && twdl_width <= 24) return 2;
    return 0; // (the per-stage conditions are checked by the planner on the stage list)
}

hipError_t launch_widelong(int log2n, const WideArgs &a, int in_cb, const void *in, void *out, void *scratch, const int2 *tw_all, const int2 *h_tw,
                           size_t nframes, hipStream_t stream, int direction)
{
    if (nframes == 0) return hipSuccess;
    W2Consts k;
    for (int i = 0; i < 8; ++i) k.wr3[i] = h_tw[7 + i].x, k.wi3[i] = h_tw[7 + i].y;
    for (int i = 0; i < 4; ++i) k.wr2[i] = h_tw[3 + i].x, k.wi2[i] = h_tw[3 + i].y;
    const bool in16 = in_cb == 2;
    if (direction == 1) {
        switch (log2n) {
        case 17: return launch_long_inv<1>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
        case 18: return launch_long_inv<2>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
        case 19: return launch_long_inv<3>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
        default: return launch_long_inv<4>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
        }
    }
    switch (log2n) {
    case 17: return launch_long<1>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
    case 18: return launch_long<2>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
    case 19: return launch_long<3>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
    default: return launch_long<4>(a, k, in16, in, out, scratch, tw_all, nframes, stream);
    }
}

} // namespace intfft
