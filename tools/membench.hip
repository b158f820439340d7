// membench.hip -- access-pattern ceilings for the N=1024 wave kernel (diagnostic tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_x4(const uint4* in, uint4* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < n; i += st) out[i] = in[i];
}
// one wave = one 4 KiB frame, 16 dword loads then 16 dword stores (the fast kernel's pattern)
template <int NT, int REVSTORE>
__global__ __launch_bounds__(256) void copy_wave_dw(const u32* in, u32* out, size_t nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t f = (size_t)blockIdx.x * 4 + wv; f < nframes; f += (size_t)gridDim.x * 4) {
        const u32* s = in + f * 1024 + lane;
        u32* d = out + f * 1024 + lane;
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = REVSTORE ? (((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3)) : j;
            if (NT) __builtin_nontemporal_store(v[j], d + 64 * r); else d[64 * r] = v[j];
        }
    }
}
// one wave = one frame, lane holds 4 x uint4 (x4 loads at 1 KiB stride), stores same
template <int NT>
__global__ __launch_bounds__(256) void copy_wave_x4(const uint4* in, uint4* out, size_t nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t f = (size_t)blockIdx.x * 4 + wv; f < nframes; f += (size_t)gridDim.x * 4) {
        const uint4* s = in + f * 256 + lane;
        uint4* d = out + f * 256 + lane;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (NT) { v4u t = __builtin_nontemporal_load((const v4u*)(s + 64 * j)); v[j] = make_uint4(t.x, t.y, t.z, t.w); }
            else v[j] = s[64 * j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (NT) { v4u t = {v[j].x, v[j].y, v[j].z, v[j].w}; __builtin_nontemporal_store(t, (v4u*)(d + 64 * j)); }
            else d[64 * j] = v[j];
        }
    }
}
// lane holds 16 consecutive dwords (64 B), 4 x uint4 at 64-B lane stride (BITREV-out store pattern)
__global__ __launch_bounds__(256) void copy_wave_lane64(const uint4* in, uint4* out, size_t nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t f = (size_t)blockIdx.x * 4 + wv; f < nframes; f += (size_t)gridDim.x * 4) {
        const uint4* s = in + f * 256 + lane * 4;
        uint4* d = out + f * 256 + lane * 4;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = s[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = v[j];
    }
}

template <typename F> float timeit(F f, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main(int argc, char** argv)
{
    size_t nframes = argc > 1 ? strtoull(argv[1], 0, 0) : 65536;
    int blocks_per_cu = argc > 2 ? atoi(argv[2]) : 8;
    size_t bytes = nframes * 4096;
    void *in, *out;
    hipMalloc(&in, bytes); hipMalloc(&out, bytes);
    hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    int cus; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned grid = cus * blocks_per_cu;
    auto rep = [&](const char* name, float ms) {
        printf("%-28s frames=%zu grid=%u  %.3f ms  %.1f GB/s (r+w)\n", name, nframes, grid, ms, 2.0 * bytes / ms / 1e6);
    };
    rep("copy_x4", timeit([&] { hipLaunchKernelGGL(copy_x4, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, bytes / 16); }, 20));
    rep("wave_dw", timeit([&] { hipLaunchKernelGGL((copy_wave_dw<0,0>), dim3(grid), dim3(256), 0, 0, (const u32*)in, (u32*)out, nframes); }, 20));
    rep("wave_dw_revstore", timeit([&] { hipLaunchKernelGGL((copy_wave_dw<0,1>), dim3(grid), dim3(256), 0, 0, (const u32*)in, (u32*)out, nframes); }, 20));
    rep("wave_dw_nt", timeit([&] { hipLaunchKernelGGL((copy_wave_dw<1,0>), dim3(grid), dim3(256), 0, 0, (const u32*)in, (u32*)out, nframes); }, 20));
    rep("wave_x4", timeit([&] { hipLaunchKernelGGL((copy_wave_x4<0>), dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, nframes); }, 20));
    rep("wave_x4_nt", timeit([&] { hipLaunchKernelGGL((copy_wave_x4<1>), dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, nframes); }, 20));
    rep("wave_lane64", timeit([&] { hipLaunchKernelGGL(copy_wave_lane64, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, nframes); }, 20));
    // sustained behaviour of the NT wave copy: chunks of 200 launches
    for (int c = 0; c < (argc > 3 ? atoi(argv[3]) : 0); ++c)
        rep("wave_dw_nt sustained", timeit([&] { hipLaunchKernelGGL((copy_wave_dw<1,0>), dim3(grid), dim3(256), 0, 0, (const u32*)in, (u32*)out, nframes); }, 200));
    return 0;
}
