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
#include "../../include/intfft.h"
#include "intfft_internal.hpp"

#include <algorithm>

namespace intfft {

struct ReorderArgs {
    int L, U;
    // tile-local bit k (0 <= k < U) of the STORE enumeration sits at m_out bit out_pos[k] (out_pos[k] = k for k < TB);
    // tile-local bit k of the LOAD enumeration sits at m_out bit ld_pos[k] (the bits feeding m_in bits 0..TB-1 first);
    // st_of_ld[k]: the store-enumeration bit of load-enumeration bit k (LDS address of a loaded element)
    signed char out_pos[12], ld_pos[12], st_of_ld[12];
    signed char rest_pos[20]; // the L - U m_out bits numbered by the block index, ascending
    signed char in_of_out[20]; // m_out bit b lands at m_in bit in_of_out[b]
};

template <typename E> __global__ __launch_bounds__(256) void k_reorder(const E *in, E *out, const ReorderArgs a, unsigned tiles_per_frame)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    E *lds = reinterpret_cast<E *>(lds_raw);
    const size_t frame = blockIdx.x / tiles_per_frame;
    const unsigned tile = blockIdx.x % tiles_per_frame;
    unsigned base_out = 0, base_in = 0; // contribution of the block-numbered bits
    for (int k = 0; k < a.L - a.U; ++k)
        if ((tile >> k) & 1u) {
            base_out |= 1u << a.rest_pos[k];
            base_in |= 1u << a.in_of_out[(int)a.rest_pos[k]];
        }
    const E *src = in + (frame << a.L);
    E *dst = out + (frame << a.L);
    const unsigned n = 1u << a.U;
    for (unsigned e = threadIdx.x; e < n; e += 256) { // consecutive e -> consecutive m_in
        unsigned m_in = base_in, slot = 0;
        for (int k = 0; k < a.U; ++k)
            if ((e >> k) & 1u) {
                m_in |= 1u << a.in_of_out[(int)a.ld_pos[k]];
                slot |= 1u << a.st_of_ld[k];
            }
        lds[slot + (slot >> 5)] = src[m_in]; // + slot / 32: breaks the power-of-two bank stride of the transposing writes
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < n; e += 256) { // consecutive e -> consecutive m_out
        unsigned m_out = base_out;
        for (int k = 0; k < a.U; ++k)
            if ((e >> k) & 1u) m_out |= 1u << a.out_pos[k];
        dst[m_out] = lds[e + (e >> 5)];
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

extern "C" int intfft_reorder(int log2n, int container_bytes, int from_order, int to_order, const void *d_in, void *d_out,
                              size_t batch, int hip_device, void *hip_stream)
{
    if (log2n < 3 || log2n > 20) return INTFFT_ERR_INVALID;
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

    ReorderArgs a{};
    a.L = L;
    // logical bit j: m_in bit mem_bit(from, j) = m_out bit mem_bit(to, j)
    int out_of_in[20];
    for (int j = 0; j < L; ++j) {
        const int bi = mem_bit_of_logical(from_order, L, j), bo = mem_bit_of_logical(to_order, L, j);
        a.in_of_out[bo] = (signed char)bi;
        out_of_in[bi] = bo;
    }
    const int TB = std::min(6, L); // 64-element runs: 256 B of int16 pairs
    bool in_tile[20] = {false};
    int U = 0;
    for (int k = 0; k < TB; ++k) { // store enumeration: m_out bits 0..TB-1 first
        a.out_pos[U++] = (signed char)k;
        in_tile[k] = true;
    }
    for (int k = 0; k < TB; ++k) { // then the m_out bits that feed m_in bits 0..TB-1
        const int b = out_of_in[k];
        if (!in_tile[b]) {
            a.out_pos[U++] = (signed char)b;
            in_tile[b] = true;
        }
    }
    a.U = U;
    // load enumeration: the feeders of m_in bits 0..TB-1 in that order, then the remaining tile bits
    int nl = 0;
    bool used[20] = {false};
    for (int k = 0; k < TB; ++k) {
        a.ld_pos[nl++] = (signed char)out_of_in[k];
        used[out_of_in[k]] = true;
    }
    for (int k = 0; k < U; ++k)
        if (!used[(int)a.out_pos[k]]) a.ld_pos[nl++] = a.out_pos[k];
    for (int k = 0; k < U; ++k)
        for (int q = 0; q < U; ++q)
            if (a.out_pos[q] == a.ld_pos[k]) a.st_of_ld[k] = (signed char)q;
    int nr = 0;
    for (int b = 0; b < L; ++b)
        if (!in_tile[b]) a.rest_pos[nr++] = (signed char)b;

    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    hipError_t e;
    switch (container_bytes) {
    case 2: e = launch<uint32_t>(a, d_in, d_out, batch, stream); break;
    case 4: e = launch<uint2>(a, d_in, d_out, batch, stream); break;
    default: e = launch<uint4>(a, d_in, d_out, batch, stream); break;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return (int)e;
}
