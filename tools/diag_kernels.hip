// diag_kernels.hip -- on-box ceilings printed next to the bench line (libintfft_diag.so; NOT part of the product library
// and never on the measured path).  Two probes, both launched by bench.py, which does the timing with HIP events:
//   diag_copy_wave_ld   a copy with the access pattern of k_fft1024_i16 (one wave per 4 KiB frame, 16 plain dword loads then 16
//                       non-temporal dword stores per lane): the memory-side ceiling of that kernel's own pattern
//   diag_copy_wave_nt   the same with non-temporal loads as well (the kernel's pattern in rounds 1-2)
//   diag_xcc_map        the XCD (HW_REG_XCC_ID) every workgroup of a launch ran on: the half-line tiles of intfft_big2x.hip pair blocks b and
//                       b + 8 and rely on the dispatcher's round robin putting them on one XCD (tests/test_gpu_xcd.py fails loudly if not)
//   diag_valu_chain     N dependent-free packed-int16 adds per lane on 8 register sets: the issue rate of the "slow class"
//                       VALU instructions (v_pk_*, v_dot2, v_perm, v_bfe) the packed butterflies are made of
//   diag_body           round 6: the butterfly BODIES of the packed 16-bit kernels (the group4 / group4_dit templates of
//                       intfftk_amd/csrc/intfft_pk16.hpp, included here as they are) run on registers only -- no global, no LDS traffic
//                       inside the timed loop: wave-rounds per second of "four general radix-2 stages on 16 registers" (32 butterflies
//                       per lane) and of "stages 3, 2 (wave-uniform twiddles), 1, 0" per arithmetic mode.  bench.py turns them into the
//                       VALU-issue floor of a mode: an N = 1024 frame is 1.5 rounds of the first kind + 1 of the second per wave.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../intfftk_amd/csrc/intfft_pk16.hpp"
typedef unsigned u32;

// NTLD: loads non-temporal too (the pattern of rounds 1-2); 0: plain loads + non-temporal stores, what the kernel does since round 3
template <int NTLD> __global__ __launch_bounds__(256) void k_copy_wave_nt(const u32 *in, u32 *out, size_t nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t f = (size_t)blockIdx.x * 4 + wv; f < nframes; f += (size_t)gridDim.x * 4) {
        const u32 *s = in + f * 1024 + lane;
        u32 *d = out + f * 1024 + lane;
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = NTLD ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
#pragma unroll
        for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j], d + 64 * j);
    }
}

// out[b] = XCC id of block b; the shape of k_big2x_a (512 threads, 68 KiB of LDS: two workgroups per CU), a short spin so that the
// launch outlives its own dispatch (blocks of a trivially short kernel would all fit the first CUs that come free)
__global__ __launch_bounds__(512) void k_xcc_map(u32 *out, int spin)
{
    extern __shared__ u32 lds_dummy[];
    u32 acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) {
        lds_dummy[threadIdx.x] = acc;
        __syncthreads();
        acc += lds_dummy[(threadIdx.x + 1) & 511];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u) | (acc == 0xffffffffu ? 16u : 0u);
}

#define DIAG_UNROLL 8
template <int SLOW> __global__ __launch_bounds__(256) void k_valu_chain(u32 *out, u32 seed, int iters)
{
    u32 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const u32 b = a0 ^ 0x5a5au;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < DIAG_UNROLL; ++u) {
            if (SLOW)
                asm volatile("v_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\t"
                             "v_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b));
            else
                asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                             "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b));
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// ---- butterfly bodies on registers ------------------------------------------------------------------------------------------
// KIND 0: four general DIF stages (register offsets 8, 4, 2, 1; per-lane twiddles) -- ROUND 0 with FASTX 1 (fast extraction), 2 (t = 16
//         exact extraction), 0 (v_bfe extraction); ROUND 1 = RNDMODE 1 on 16-bit data
// KIND 1: DIF stages 3, 2 (wave-uniform twiddles) + STAGE 1 + STAGE 0 (dif_round_c; FASTX 2 runs its general stages through group4<.., 2, ..>)
// KIND 2 / 3: the DIT mirrors (dit_round with the DIT packing / dit_round_c)
template <int ROUND, int FASTX> __device__ __forceinline__ void body_dif4(u32 (&v)[16], const intfft::RoundTw &tw, const intfft::Slice &sl)
{
    using namespace intfft;
    constexpr bool P = ROUND == 0;
    constexpr int MA = P ? 0xF : 0;
    const u32 wa0[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[2], tw.wa8[3]}, wb0[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[2], tw.wb8[3]};
    const u32 wa1[4] = {tw.wa8[4], tw.wa8[5], tw.wa8[6], tw.wa8[7]}, wb1[4] = {tw.wb8[4], tw.wb8[5], tw.wb8[6], tw.wb8[7]};
    group4<ROUND, FASTX, false, P, false, 0>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
    group4<ROUND, FASTX, false, P, false, 0>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
    group4<ROUND, FASTX, false, P, false, 0>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], tw.wa4, tw.wb4, sl);
    group4<ROUND, FASTX, false, P, false, MA>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], tw.wa4, tw.wb4, sl);
    const u32 wa2[4] = {tw.wa2[0], tw.wa2[1], tw.wa2[0], tw.wa2[1]}, wb2[4] = {tw.wb2[0], tw.wb2[1], tw.wb2[0], tw.wb2[1]};
    group4<ROUND, FASTX, false, P, false, 0>(v[0], v[2], v[1], v[3], v[8], v[10], v[9], v[11], wa2, wb2, sl);
    group4<ROUND, FASTX, false, P, false, MA>(v[4], v[6], v[5], v[7], v[12], v[14], v[13], v[15], wa2, wb2, sl);
    const u32 wa3[4] = {tw.wa1[0], tw.wa1[0], tw.wa1[0], tw.wa1[0]}, wb3[4] = {tw.wb1[0], tw.wb1[0], tw.wb1[0], tw.wb1[0]};
    group4<ROUND, FASTX, false, P, false, 0>(v[0], v[1], v[4], v[5], v[8], v[9], v[12], v[13], wa3, wb3, sl);
    group4<ROUND, FASTX, false, P, false, MA>(v[2], v[3], v[6], v[7], v[10], v[11], v[14], v[15], wa3, wb3, sl);
}
template <int ROUND, int FASTX> __device__ __forceinline__ void body_difc(u32 (&v)[16], const intfft::RoundCConsts &c, const intfft::Slice &sl)
{
    using namespace intfft;
    if constexpr (FASTX != 2) {
        dif_round_c<FASTX == 1, ROUND>(v, c, sl, v2s{0, 0});
    } else { // the t = 16 exact extraction on the two general stages, then STAGE 1 / 0 as dif_round_c has them
        const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
        const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
        group4<0, 2, false, true, true, 0, true>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl, v2s{1, 1});
        group4<0, 2, false, true, true, 0, true>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl, v2s{1, 1});
        group4<0, 2, false, true, true, 0>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
        group4<0, 2, false, true, true, 0xF>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
#pragma unroll
        for (int g = 0; g < 16; g += 8) {
            bfly_triv<false, false>(v[g], v[g + 2]);
            bfly_mj<false, false>(v[g + 1], v[g + 3]);
            bfly_triv<false, true>(v[g + 4], v[g + 6]);
            bfly_mj<false, true>(v[g + 5], v[g + 7]);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) bfly_triv<false, false>(v[g], v[g + 1]);
    }
}

template <int KIND, int ROUND, int FASTX>
__global__ __launch_bounds__(256) void k_body(const u32 *seed, u32 *out, const intfft::RoundCConsts c, const intfft::Slice sl, int iters)
{
    using namespace intfft;
    u32 v[16];
    RoundTw tw;
    const u32 *s = seed + threadIdx.x * 48;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = s[j];
    u32 *twp = reinterpret_cast<u32 *>(&tw);
#pragma unroll
    for (int j = 0; j < 30; ++j) twp[j] = s[16 + j];
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) body_dif4<ROUND, FASTX>(v, tw, sl);
        else if constexpr (KIND == 1) body_difc<ROUND, FASTX>(v, c, sl);
        else if constexpr (KIND == 2) dit_round<FASTX == 1, 4, ROUND, true>(v, tw, sl);
        else dit_round_c<FASTX == 1, ROUND, true>(v, c, sl);
    }
    u32 acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= v[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

extern "C" {

// kind 0..3 (see above), round 0 / 1, fastx 0 / 1 / 2 (DIT kinds: 0 / 1).  d_seed: >= 256 * 48 dwords of arbitrary data, d_out: >= 4 * CUs * 256
// dwords.  4 waves per SIMD on every CU, like the wave kernels.  *n_wave_rounds = wave-rounds executed (one round = 32 butterflies per lane).
// returns a hipError_t, or -1 for a combination that is not instantiated
int diag_body(int kind, int round, int fastx, int iters, const void *d_seed, void *d_out, unsigned long long *n_wave_rounds, void *stream)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)cus * 4;
    intfft::RoundCConsts c;
    for (int k = 0; k < 8; ++k) c.wa3[k] = 0x12345678u * (k + 1), c.wb3[k] = 0x9abcdef1u * (k + 3);
    for (int k = 0; k < 4; ++k) c.wa2[k] = 0x0f1e2d3cu * (k + 5), c.wb2[k] = 0x4b5a6978u * (k + 7);
    intfft::Slice sl{15, 16, 0x05040100u, 0x07060302u};
#define DIAG_BODY(K, R, F)                                                                                                      \
    if (kind == K && round == R && fastx == F) {                                                                                 \
        hipLaunchKernelGGL((k_body<K, R, F>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32 *)d_seed, (u32 *)d_out, c, sl, iters); \
        if (n_wave_rounds) *n_wave_rounds = (unsigned long long)grid * 4ull * (unsigned long long)iters;                         \
        return (int)hipGetLastError();                                                                                           \
    }
    DIAG_BODY(0, 0, 1) DIAG_BODY(0, 0, 2) DIAG_BODY(0, 0, 0) DIAG_BODY(0, 1, 0)
    DIAG_BODY(1, 0, 1) DIAG_BODY(1, 0, 2) DIAG_BODY(1, 0, 0) DIAG_BODY(1, 1, 0)
    DIAG_BODY(2, 0, 1) DIAG_BODY(2, 0, 0) DIAG_BODY(2, 1, 0)
    DIAG_BODY(3, 0, 1) DIAG_BODY(3, 0, 0) DIAG_BODY(3, 1, 0)
#undef DIAG_BODY
    return -1;
}

// grid = blocks_per_cu x CUs of the current device; returns a hipError_t
int diag_copy_wave_nt(const void *d_in, void *d_out, size_t nframes, int blocks_per_cu, void *stream)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(k_copy_wave_nt<1>, dim3((unsigned)(cus * blocks_per_cu)), dim3(256), 0, (hipStream_t)stream,
                       (const u32 *)d_in, (u32 *)d_out, nframes);
    return (int)hipGetLastError();
}
// the same with plain loads (non-temporal stores only)
int diag_copy_wave_ld(const void *d_in, void *d_out, size_t nframes, int blocks_per_cu, void *stream)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(k_copy_wave_nt<0>, dim3((unsigned)(cus * blocks_per_cu)), dim3(256), 0, (hipStream_t)stream,
                       (const u32 *)d_in, (u32 *)d_out, nframes);
    return (int)hipGetLastError();
}

// launches 4 waves per SIMD on every CU (grid = 4 x CUs blocks of 256); d_out holds >= 4 * CUs * 256 dwords.
// wave-instructions issued per launch = *n_wave_insts (8 * DIAG_UNROLL * iters per wave)
int diag_valu_chain(int slow, int iters, void *d_out, unsigned long long *n_wave_insts, void *stream)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)cus * 4;
    if (slow) hipLaunchKernelGGL(k_valu_chain<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (u32 *)d_out, 12345u, iters);
    else hipLaunchKernelGGL(k_valu_chain<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (u32 *)d_out, 12345u, iters);
    if (n_wave_insts) *n_wave_insts = (unsigned long long)grid * 4ull * 8ull * DIAG_UNROLL * (unsigned long long)iters;
    return (int)hipGetLastError();
}

// d_out: nblocks dwords.  returns a hipError_t
int diag_xcc_map(void *d_out, unsigned nblocks, int spin, void *stream)
{
    hipError_t e = hipFuncSetAttribute((const void *)k_xcc_map, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_xcc_map, dim3(nblocks), dim3(512), 68 * 1024, (hipStream_t)stream, (u32 *)d_out, spin);
    return (int)hipGetLastError();
}

int diag_device_cus(void)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}
}
