// diag_kernels.hip -- on-box ceilings printed next to the bench line (libintfft_diag.so; NOT part of the product library
// and never on the measured path).  Two probes, both launched by bench.py, which does the timing with HIP events:
//   diag_copy_wave_ld   a copy with the access pattern of k_fft1024_i16 (one wave per 4 KiB frame, 16 plain dword loads then 16
//                       non-temporal dword stores per lane): the memory-side ceiling of that kernel's own pattern
//   diag_copy_wave_nt   the same with non-temporal loads as well (the kernel's pattern in rounds 1-2)
//   diag_xcc_map        the XCD (HW_REG_XCC_ID) every workgroup of a launch ran on: the half-line tiles of intfft_big2x.hip pair blocks b and
//                       b + 8 and rely on the dispatcher's round robin putting them on one XCD (tests/test_gpu_xcd.py fails loudly if not)
//   diag_valu_chain     N dependent-free packed-int16 adds per lane on 8 register sets: the issue rate of the "slow class"
//                       VALU instructions (v_pk_*, v_dot2, v_perm, v_bfe) the packed butterflies are made of
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
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

extern "C" {

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
