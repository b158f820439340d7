// intfft_fast4096w.hip -- general-width block kernel: int_fftNk for N = 2048 / 4096 with any DATA_WIDTH / TWDL_WIDTH /
// FORMAT / RNDMODE whose widths stay within 32 bits (e.g. the unscaled 16-bit transform: 28-bit results), natural
// order in and out.  (DATA_WIDTH = 16 scaled-truncate with TWDL_WIDTH <= 16 has the packed kernel intfft_fast4096.hip.)
//
// Workgroup mapping of intfft_fast4096.hip -- 256 threads own 4096 consecutive samples (one frame, or two 2048-point
// frames whose frame-number stage is skipped), 16 samples per thread, three register rounds of four stages:
//   LA  reg = n11..8, thread = n7..0            STAGE 11..8
//   LB  reg = n7..4,  thread = (n11..8, n3..0)  STAGE 7..4
//   LC  reg = n3..0,  thread = lc_bit<L>()      STAGE 3..0, stored with the bit reversal folded into the mapping
// with block-wide LDS transposes of two dword planes (re, im) in one 40 KiB region, on unpacked int32 registers with
// the parameterised butterflies of intfft_u32.hpp (gfly: every multiplier regime, trunc / round / unscaled).
// Thread-dependent twiddles (30 pairs) are frame invariant and live in VGPRs.
// rhu2 sums in their long form here: with the short form of intfft_u32.hpp (four operations for rhu2(A + B), one more for rhu2(A - B)) the
// scheduler interleaves more butterflies and the 4096-point block kernels spill under their 128-register bound (60 .. 132 bytes per lane
// of scratch in the RNDMODE = 1 instantiations, 178 against 199 Gsample/s measured); every other kernel family takes the short form.
#define INTFFT_RHU2_LONG
#include "intfft_u32.hpp"

namespace intfft {

constexpr int ROW4W = 20;
constexpr int PLANE4W = 256 * ROW4W;

template <int L> __host__ __device__ constexpr int lcw_bit(int k) { return k < L ? (L - 1) - k : (L - 4) + (k - L); }
template <int L> __host__ __device__ constexpr int lcw_row_of_reg(int j)
{
    return (((j >> 0) & 1) << lcw_bit<L>(4)) | (((j >> 1) & 1) << lcw_bit<L>(5)) | (((j >> 2) & 1) << lcw_bit<L>(6)) |
           (((j >> 3) & 1) << lcw_bit<L>(7));
}
__device__ __forceinline__ constexpr int rev4q(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// one register round: stages s0+3 .. s0 on register offsets 8, 4, 2, 1 (only those below L)
template <int MODE, bool MASKED, int L, int S0>
__device__ __forceinline__ void ground(int (&re)[16], int (&im)[16], const int (&w8r)[8], const int (&w8i)[8],
                                       const int (&w4r)[4], const int (&w4i)[4], const int (&w2r)[2], const int (&w2i)[2],
                                       int w1r, int w1i, const W32Args &a)
{
    if constexpr (S0 + 3 < L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gfly<MODE, false, MASKED>(re[j], im[j], re[j + 8], im[j + 8], w8r[j], w8i[j], a.st[S0 + 3]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r[j], w4i[j], a.st[S0 + 2]);
#pragma unroll
    for (int g = 0; g < 16; g += 4)
#pragma unroll
        for (int j = 0; j < 2; ++j) gfly<MODE, false, MASKED>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w2r[j], w2i[j], a.st[S0 + 1]);
#pragma unroll
    for (int g = 0; g < 16; g += 2) gfly<MODE, false, MASKED>(re[g], im[g], re[g + 1], im[g + 1], w1r, w1i, a.st[S0]);
}

// OUT64 (unscaled only): 0 = results within 32 bits; 1 = 33 / 34-bit results (stages 1, 0 in 64 bits); 2 = up to 40 bits (the
// whole last round in 64 bits).  A template parameter, not a runtime branch: the 64-bit rounds hold 64 more live dwords, and
// compiled into the 32-bit kernel they cost it 120 spilled dwords per lane under the 128-VGPR cap of four waves per SIMD
// (16-bit unscaled N = 4096: 139 Gsample/s).  The 64-bit variants run three waves per SIMD instead.
// NAT (round 4): the instantiation for int_fftNk's own beat orders (`native` bit 0: HALVES in, bit 1: BITREV out; results within 32 bits): HALVES
// beats = one 8- / 16-byte load of the LA register pair (j0, j0 | 2^(L-9)); BITREV order = the core position, 16 consecutive ones per LC thread,
// through the (then idle) transpose region in memory order -- padded rows of 16 samples -- 1 KiB per wave instruction
template <int L, int MODE, bool MASKED, int OUT64, bool NAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OUT64 ? 3 : 4, OUT64 ? 3 : 4)))
void k_fft4096_w32(const void *in, void *out, const int2 *__restrict__ twt, const UConsts c, const W32Args a,
                   size_t nframes_user, int native)
{
    static_assert(!NAT || OUT64 == 0, "native beat orders: int16 / int32 results");
    const bool halves = NAT && (native & 1), bitrev = NAT && (native & 2);
    static_assert(L == 11 || L == 12, "block kernel: N = 2048 or 4096");
    constexpr int FP = 1 << (12 - L);
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks of 4096 samples
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANE4W];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;

    // frame-invariant twiddles: round A (thread = n7..0): STAGE 11 index 256 jj + tid .. STAGE 8 index tid;
    // round B (thread low nibble = n3..0): STAGE 7 index 16 jj + lo4 .. STAGE 4 index lo4
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
    // LA -> LB: element (thread x, reg y) -> row 16 y + x3..0, column x7..4
    u32 *const w_ab = lds + ROW4W * lo4 + hi4;
    // LB -> LC: thread (hi4 = n11..8, lo4 = n3..0), reg j' = n7..4 -> row = LC thread (lcw_bit<L>), column n3..0
    const int row_hi = ((hi4 & 1) << lcw_bit<L>(8)) | (((hi4 >> 1) & 1) << lcw_bit<L>(9)) | (((hi4 >> 2) & 1) << lcw_bit<L>(10)) |
                       (((hi4 >> 3) & 1) << lcw_bit<L>(11));
    u32 *const w_bc = lds + ROW4W * row_hi + lo4;
    const uint4 *const rd0 = reinterpret_cast<const uint4 *>(lds + ROW4W * tid);
    const uint4 *const rd1 = reinterpret_cast<const uint4 *>(lds + PLANE4W + ROW4W * tid);
    // LC <-> natural-order X: index = rev4(r) * 2^(L-4) + lc_off
    int lc_off = 0, lc_frame = 0;
#pragma unroll
    for (int k = 4; k < 12; ++k) {
        const int bit = (tid >> lcw_bit<L>(k)) & 1;
        lc_off += bit * (k >= L ? (1 << k) : (1 << (L - 1 - k)));
        if (k >= L) lc_frame += bit << (k - L);
    }

    auto transpose_read = [&](int (&re)[16], int (&im)[16]) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd0[q], y = rd1[q];
            re[4 * q + 0] = (int)x.x, re[4 * q + 1] = (int)x.y, re[4 * q + 2] = (int)x.z, re[4 * q + 3] = (int)x.w;
            im[4 * q + 0] = (int)y.x, im[4 * q + 1] = (int)y.y, im[4 * q + 2] = (int)y.z, im[4 * q + 3] = (int)y.w;
        }
        __syncthreads(); // the region is rewritten by the next transpose
    };

    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const bool partial = L < 12 && (f + 1) * FP > nframes_user; // last chunk: the absent frame reads as 0
        int re[16], im[16];
        // (wave-uniform pointer)[32-bit thread offset] for every global access (at32, intfft_device.hpp), the partial-chunk test around the loops:
        // per-access 64-bit addresses and a test inside the unrolled loads cost these kernels 4-24 spilled VGPRs and serialised loads (round 4)
        unsigned tl = (unsigned)tid, lco = (unsigned)lc_off;
        asm volatile("" : "+v"(tl), "+v"(lco));
        if (NAT && halves) {
            constexpr int HB = 1 << (L - 9); // LA register bit that carries n(L-1)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j0 = ((jj / HB) * 2 * HB) | (jj % HB);
                const int p0 = 256 * j0; // logical position P = p0 + tid (bit L-1 clear): frame P >> L, beat P mod N/2
                const int pair = ((p0 >> L) << (L - 1)) | (p0 & ((1 << (L - 1)) - 1));
                const bool ok = !partial || f * FP + (size_t)(p0 >> L) < nframes_user;
                if (a.in16) {
                    typedef u32 v2u __attribute__((ext_vector_type(2)));
                    v2u w = {0u, 0u};
                    if (ok) w = INTFFT_LD(at32(reinterpret_cast<const v2u *>(static_cast<const u32 *>(in) + f * 4096) + pair, tl));
                    re[j0] = (int)(w.x << a.in_sh) >> a.in_sh, im[j0] = (int)(w.x << (a.in_sh - 16)) >> a.in_sh;
                    re[j0 | HB] = (int)(w.y << a.in_sh) >> a.in_sh, im[j0 | HB] = (int)(w.y << (a.in_sh - 16)) >> a.in_sh;
                } else {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    v4i w = {0, 0, 0, 0};
                    if (ok) w = INTFFT_LD(at32(reinterpret_cast<const v4i *>(static_cast<const int2 *>(in) + f * 4096) + pair, tl));
                    re[j0] = (int)((u32)w.x << a.in_sh) >> a.in_sh, im[j0] = (int)((u32)w.y << a.in_sh) >> a.in_sh;
                    re[j0 | HB] = (int)((u32)w.z << a.in_sh) >> a.in_sh, im[j0 | HB] = (int)((u32)w.w << a.in_sh) >> a.in_sh;
                }
            }
        } else if (a.in16) {
            const u32 *src = static_cast<const u32 *>(in) + f * 4096; // wave-uniform
            u32 raw[16];
            if (!partial) {
#pragma unroll
                for (int j = 0; j < 16; ++j) raw[j] = INTFFT_LD(at32(src + 256 * j, tl));
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) raw[j] = f * FP + (size_t)((256 * j + tid) >> L) < nframes_user ? INTFFT_LD(at32(src + 256 * j, tl)) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = (int)(raw[j] << a.in_sh) >> a.in_sh;
                im[j] = (int)(raw[j] << (a.in_sh - 16)) >> a.in_sh;
            }
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i *src = static_cast<const v2i *>(in) + f * 4096;
            v2i x[16];
            if (!partial) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = INTFFT_LD(at32(src + 256 * j, tl));
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x[j] = v2i{0, 0};
                    if (f * FP + (size_t)((256 * j + tid) >> L) < nframes_user) x[j] = INTFFT_LD(at32(src + 256 * j, tl));
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = (int)((u32)x[j].x << a.in_sh) >> a.in_sh;
                im[j] = (int)((u32)x[j].y << a.in_sh) >> a.in_sh;
            }
        }
        ground<MODE, MASKED, L, 8>(re, im, a8r, a8i, a4r, a4i, a2r, a2i, a1r, a1i, a);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            w_ab[ROW4W * 16 * j] = (u32)re[j];
            w_ab[PLANE4W + ROW4W * 16 * j] = (u32)im[j];
        }
        transpose_read(re, im);
        ground<MODE, MASKED, 12, 4>(re, im, b8r, b8i, b4r, b4i, b2r, b2i, b1r, b1i, a);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            w_bc[ROW4W * lcw_row_of_reg<L>(j)] = (u32)re[j];
            w_bc[PLANE4W + ROW4W * lcw_row_of_reg<L>(j)] = (u32)im[j];
        }
        transpose_read(re, im);
        if constexpr (MODE == W_UNSCALED && OUT64 == 2) { // 35 / 36-bit results: the whole round LC in 64 bits (gfly64, intfft_u32.hpp)
            long long xr[16], xi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[r] = re[r], xi[r] = im[r];
#pragma unroll
            for (int r = 0; r < 8; ++r) gfly64<MASKED>(xr[r], xi[r], xr[r + 8], xi[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) gfly64<MASKED>(xr[g + r], xi[g + r], xr[g + r + 4], xi[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
            tail64_stages10(xr, xi);
            if (!partial || f * FP + (size_t)lc_frame < nframes_user) {
                typedef long long v2l __attribute__((ext_vector_type(2)));
                v2l *dst = reinterpret_cast<v2l *>(out) + f * 4096; // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const v2l y = {xr[r], xi[r]};
                    __builtin_nontemporal_store(y, at32(dst + (rev4q(r) << (L - 4)), lco));
                }
            }
            continue;
        }
        // LC: stages 3, 2 (uniform twiddles), 1, 0
#pragma unroll
        for (int r = 0; r < 8; ++r) gfly<MODE, true, MASKED>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], a.st[3]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) gfly<MODE, true, MASKED>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], a.st[2]);
        if constexpr (MODE == W_UNSCALED && OUT64 == 1) { // 33 / 34-bit results: stages 1, 0 in 64 bits, int64 containers
            long long xr[16], xi[16];
            tail64_unscaled(re, im, xr, xi);
            if (!partial || f * FP + (size_t)lc_frame < nframes_user) {
                typedef long long v2l __attribute__((ext_vector_type(2)));
                v2l *dst = reinterpret_cast<v2l *>(out) + f * 4096; // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const v2l y = {xr[r], xi[r]};
                    __builtin_nontemporal_store(y, at32(dst + (rev4q(r) << (L - 4)), lco));
                }
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            gfly_triv<MODE, false>(re[g], im[g], re[g + 2], im[g + 2], a.st[1]);
            gfly_triv<MODE, true>(re[g + 1], im[g + 1], re[g + 3], im[g + 3], a.st[1]);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) gfly_triv<MODE, false>(re[g], im[g], re[g + 1], im[g + 1], a.st[0]);

        if (NAT && bitrev) {
            // position of (thread, reg r) = A(tid) | r: 16 consecutive samples per thread -> one padded row of the transpose region (free here:
            // the last transpose_read ends with a barrier)
            typedef int v4i __attribute__((ext_vector_type(4)));
            int A = 0;
#pragma unroll
            for (int k = 4; k < 12; ++k) A |= ((tid >> lcw_bit<L>(k)) & 1) << k;
            if (a.out16) {
                typedef u32 v4u __attribute__((ext_vector_type(4)));
                u32 *row = lds + 20 * (A >> 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4u y;
                    y.x = ((u32)re[4 * q] & 0xFFFFu) | ((u32)im[4 * q] << 16), y.y = ((u32)re[4 * q + 1] & 0xFFFFu) | ((u32)im[4 * q + 1] << 16);
                    y.z = ((u32)re[4 * q + 2] & 0xFFFFu) | ((u32)im[4 * q + 2] << 16), y.w = ((u32)re[4 * q + 3] & 0xFFFFu) | ((u32)im[4 * q + 3] << 16);
                    *reinterpret_cast<v4u *>(row + 4 * q) = y;
                }
                __syncthreads();
                v4i *dst4 = static_cast<v4i *>(out) + f * 1024; // wave-uniform (4096 samples x 4 B)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 256 * i + tid; // 16-byte piece = samples 4 e .. 4 e + 3
                    if (partial && f * FP + (size_t)((4 * e) >> L) >= nframes_user) continue;
                    __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 20 * (e >> 2) + 4 * (e & 3)), at32(dst4 + 256 * i, tl));
                }
            } else {
                u32 *row = lds + 36 * (A >> 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) *reinterpret_cast<v4i *>(row + 4 * q) = v4i{re[2 * q], im[2 * q], re[2 * q + 1], im[2 * q + 1]};
                __syncthreads();
                v4i *dst4 = static_cast<v4i *>(out) + f * 2048; // wave-uniform (4096 samples x 8 B)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 256 * i + tid; // 16-byte piece = samples 2 e, 2 e + 1
                    if (partial && f * FP + (size_t)((2 * e) >> L) >= nframes_user) continue;
                    __builtin_nontemporal_store(*reinterpret_cast<const v4i *>(lds + 36 * (e >> 3) + 4 * (e & 7)), at32(dst4 + 256 * i, tl));
                }
            }
            __syncthreads(); // the region is rewritten by the next frame's transposes
        } else if (!partial || f * FP + (size_t)lc_frame < nframes_user) {
            if (a.out16) {
                u32 *dst = static_cast<u32 *>(out) + f * 4096; // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_nontemporal_store(((u32)re[r] & 0xFFFFu) | ((u32)im[r] << 16), at32(dst + (rev4q(r) << (L - 4)), lco));
            } else {
                typedef int v2i __attribute__((ext_vector_type(2)));
                v2i *dst = reinterpret_cast<v2i *>(static_cast<int2 *>(out) + f * 4096); // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const v2i y = {re[r], im[r]};
                    __builtin_nontemporal_store(y, at32(dst + (rev4q(r) << (L - 4)), lco));
                }
            }
        }
    }
}

bool fast4096w_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                         int out_order)
{
    const int out_bits = data_width + format * log2n; // unscaled: 33 / 34-bit results through the 64-bit tail stages,
    // up to 40 bits with the whole last round in 64 bits as long as STAGE 4 still fits 32 (e.g. 24-bit data: 35 / 36-bit results)
    const bool fits = out_bits <= 32 || (format == 1 && out_bits <= 34) || (format == 1 && data_width + log2n - 4 <= 32 && out_bits <= 40);
    if (!((log2n == 11 || log2n == 12) && data_width >= 2 && fits && twdl_width >= 4 && twdl_width <= 26 && direction == 0 && use_fly == 1)) return false;
    if (in_order == 0 && out_order == 0) return true;
    // int_fftNk's own beat orders (HALVES in, BITREV out) and the mixed forms: results within 32 bits
    return out_bits <= 32 && (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1);
}

const char *fast4096w_kernel_name() { return "k_fft4096_w32"; }

template <int L, int MODE, bool MASKED, int OUT64 = 0>
static hipError_t launch4w(const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a, size_t nframes,
                           hipStream_t stream, int native)
{
    const size_t chunks = (nframes + ((size_t)1 << (12 - L)) - 1) >> (12 - L);
    if constexpr (OUT64 == 0) {
        if (native) {
            const size_t capn = resident_blocks(kptr(k_fft4096_w32<L, MODE, MASKED, OUT64, true>), 256, 2);
            hipLaunchKernelGGL((k_fft4096_w32<L, MODE, MASKED, OUT64, true>), dim3((unsigned)(chunks < capn ? chunks : capn)), dim3(256), 0, stream, in, out,
                               tw, c, a, nframes, native);
            return hipGetLastError();
        }
    }
    const size_t cap = resident_blocks(kptr(k_fft4096_w32<L, MODE, MASKED, OUT64>), 256, 2);
    hipLaunchKernelGGL((k_fft4096_w32<L, MODE, MASKED, OUT64>), dim3((unsigned)(chunks < cap ? chunks : cap)), dim3(256), 0, stream, in, out,
                       tw, c, a, nframes, 0);
    return hipGetLastError();
}

template <int L>
static hipError_t launch4w_l(int mode, const void *in, void *out, const int2 *tw, const UConsts &c, const W32Args &a,
                             size_t nframes, hipStream_t stream, int native)
{
    if (a.masked) {
        switch (mode) {
        case W_TRUNC: return launch4w<L, W_TRUNC, true>(in, out, tw, c, a, nframes, stream, native);
        case W_ROUND: return launch4w<L, W_ROUND, true>(in, out, tw, c, a, nframes, stream, native);
        default:
            return a.out64 == 2   ? launch4w<L, W_UNSCALED, true, 2>(in, out, tw, c, a, nframes, stream, native)
                   : a.out64 == 1 ? launch4w<L, W_UNSCALED, true, 1>(in, out, tw, c, a, nframes, stream, native)
                                  : launch4w<L, W_UNSCALED, true>(in, out, tw, c, a, nframes, stream, native);
        }
    }
    switch (mode) {
    case W_TRUNC: return launch4w<L, W_TRUNC, false>(in, out, tw, c, a, nframes, stream, native);
    case W_ROUND: return launch4w<L, W_ROUND, false>(in, out, tw, c, a, nframes, stream, native);
    default:
        return a.out64 == 2   ? launch4w<L, W_UNSCALED, false, 2>(in, out, tw, c, a, nframes, stream, native)
               : a.out64 == 1 ? launch4w<L, W_UNSCALED, false, 1>(in, out, tw, c, a, nframes, stream, native)
                              : launch4w<L, W_UNSCALED, false>(in, out, tw, c, a, nframes, stream, native);
    }
}

hipError_t launch_fast4096w(int log2n, int mode, const W32Args &a, const void *in, void *out, const int2 *tw_all,
                            const int2 *h_tw, size_t nframes, hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    return log2n == 11 ? launch4w_l<11>(mode, in, out, tw_all, c, a, nframes, stream, native)
                       : launch4w_l<12>(mode, in, out, tw_all, c, a, nframes, stream, native);
}

} // namespace intfft
