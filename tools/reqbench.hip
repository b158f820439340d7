// tools/reqbench.hip -- calibration of the L2's memory-side request counters on gfx950 (diagnostics; not part of the product).
// Each kernel touches a KNOWN number of bytes with a known piece size, so that TCC_EA0_RDREQ / _32B / TCC_BUBBLE / WRREQ / _64B and
// FETCH_SIZE / WRITE_SIZE of the multi-pass kernels (64-byte row pieces at a 4 KiB stride) can be read in bytes instead of guessed:
//   k_rd_full        every 128-byte line whole (16 B per lane)
//   k_rd_half64      only the even 64-byte half of every line (dword loads, 16 lanes per piece)
//   k_rd_half64_odd  only the odd halves
//   k_rd_quarter32   only the first 32 bytes of every line
//   k_rd_pairs64     both halves of every line, but by DIFFERENT workgroups (blocks b and b + 8: same XCD), as the half-line tiles do
//   k_wr_*           the same shapes as stores (plain and non-temporal)
// Build: hipcc --offload-arch=gfx950 -O3 -o build/reqbench tools/reqbench.hip ; run under rocprofv3 --pmc <counters> (tools/pmc_req.sh).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

// lines = number of 128-byte lines of the buffer; grid-stride over lines
template <bool NT> __global__ __launch_bounds__(256) void k_rd_full(const v4u *in, u32 *sink, size_t lines)
{
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines * 8; i += (size_t)gridDim.x * 256) {
        const v4u x = NT ? __builtin_nontemporal_load(in + i) : in[i];
        acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
// piece: 16 lanes x 4 B = 64 B at byte offset OFF of every line
template <int OFF, bool NT> __global__ __launch_bounds__(256) void k_rd_half64(const u32 *in, u32 *sink, size_t lines)
{
    u32 acc = 0;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t p = t >> 4; p < lines; p += ((size_t)gridDim.x * 256) >> 4) {
        const u32 *q = in + p * 32 + OFF / 4 + (t & 15);
        acc ^= NT ? __builtin_nontemporal_load(q) : *q;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <bool NT> __global__ __launch_bounds__(256) void k_rd_quarter32(const u32 *in, u32 *sink, size_t lines)
{
    u32 acc = 0;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t p = t >> 3; p < lines; p += ((size_t)gridDim.x * 256) >> 3) {
        const u32 *q = in + p * 32 + (t & 7);
        acc ^= NT ? __builtin_nontemporal_load(q) : *q;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
// the tile pattern: a workgroup reads 1024 pieces of 64 B at a 4 KiB stride (one "column chunk" of a 4 MiB frame), 32 loads per thread;
// blocks b and b + 8 read the two halves of the same lines.  frames x 64 chunks blocks.
template <bool NT> __global__ __launch_bounds__(512) void k_rd_pairs64(const u32 *in, u32 *sink, size_t frames)
{
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u, G = (blockIdx.x >> 4) * 8u + slot;
    const unsigned chunk = (G & 31u) * 2u + part;
    const size_t frame = G >> 5;
    if (frame >= frames) return;
    const int tid = threadIdx.x, l = tid & 15, hx = tid >> 4;
    const u32 *src = in + (frame << 20) + ((size_t)hx << 10) + chunk * 16 + l;
    u32 acc = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const u32 *q = src + ((size_t)j << 15);
        acc ^= NT ? __builtin_nontemporal_load(q) : *q;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <bool NT> __global__ __launch_bounds__(256) void k_wr_full(v4u *out, size_t lines)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines * 8; i += (size_t)gridDim.x * 256) {
        const v4u x = {(u32)i, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(x, out + i);
        else out[i] = x;
    }
}
template <int OFF, bool NT> __global__ __launch_bounds__(256) void k_wr_half64(u32 *out, size_t lines)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t p = t >> 4; p < lines; p += ((size_t)gridDim.x * 256) >> 4) {
        u32 *q = out + p * 32 + OFF / 4 + (t & 15);
        if (NT) __builtin_nontemporal_store((u32)p, q);
        else *q = (u32)p;
    }
}
template <bool NT> __global__ __launch_bounds__(256) void k_wr_quarter32(u32 *out, size_t lines)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t p = t >> 3; p < lines; p += ((size_t)gridDim.x * 256) >> 3) {
        u32 *q = out + p * 32 + (t & 7);
        if (NT) __builtin_nontemporal_store((u32)p, q);
        else *q = (u32)p;
    }
}
// pass B's store pattern: a workgroup writes 1024 pieces of 64 B at a 4 KiB stride, the partner block (b + 8) the other halves
template <bool NT> __global__ __launch_bounds__(512) void k_wr_pairs64(u32 *out, size_t frames)
{
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u, G = (blockIdx.x >> 4) * 8u + slot;
    const unsigned chunk = (G & 31u) * 2u + part;
    const size_t frame = G >> 5;
    if (frame >= frames) return;
    const int tid = threadIdx.x, l = tid & 15, hx = tid >> 4;
    u32 *dst = out + (frame << 20) + ((size_t)hx << 10) + chunk * 16 + l;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        u32 *q = dst + ((size_t)j << 15);
        if (NT) __builtin_nontemporal_store((u32)j, q);
        else *q = (u32)j;
    }
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024; // buffer size: beyond the 256 MiB Infinity Cache by default
    const size_t bytes = mib << 20, lines = bytes / 128, frames = bytes >> 22;
    u32 *buf, *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    const unsigned tgrid = (unsigned)(frames * 64);
    for (int rep = 0; rep < 3; ++rep) {
        k_rd_full<false><<<grid, 256>>>((const v4u *)buf, sink, lines);
        k_rd_full<true><<<grid, 256>>>((const v4u *)buf, sink, lines);
        k_rd_half64<0, false><<<grid, 256>>>(buf, sink, lines);
        k_rd_half64<0, true><<<grid, 256>>>(buf, sink, lines);
        k_rd_half64<64, false><<<grid, 256>>>(buf, sink, lines);
        k_rd_quarter32<false><<<grid, 256>>>(buf, sink, lines);
        k_rd_quarter32<true><<<grid, 256>>>(buf, sink, lines);
        k_rd_pairs64<false><<<tgrid, 512>>>(buf, sink, frames);
        k_rd_pairs64<true><<<tgrid, 512>>>(buf, sink, frames);
        k_wr_full<false><<<grid, 256>>>((v4u *)buf, lines);
        k_wr_full<true><<<grid, 256>>>((v4u *)buf, lines);
        k_wr_half64<0, false><<<grid, 256>>>(buf, lines);
        k_wr_half64<0, true><<<grid, 256>>>(buf, lines);
        k_wr_half64<64, true><<<grid, 256>>>(buf, lines);
        k_wr_quarter32<false><<<grid, 256>>>(buf, lines);
        k_wr_quarter32<true><<<grid, 256>>>(buf, lines);
        k_wr_pairs64<false><<<tgrid, 512>>>(buf, frames);
        k_wr_pairs64<true><<<tgrid, 512>>>(buf, frames);
        CK(hipDeviceSynchronize());
    }
    printf("reqbench: %zu MiB, %zu lines of 128 B; full = %zu bytes, half64 = %zu, quarter32 = %zu, pairs64 = %zu\n", mib, lines, bytes, bytes / 2,
           bytes / 4, bytes);
    return 0;
}
