// tilebench.hip -- access-pattern ceilings of the multi-pass tile shapes (diagnostic tool, not product).
// A workgroup copies "tiles": ROWS rows of ROWDW dwords at a stride of `stride` dwords, read with one
// geometry and written with another (same number of elements), NREG dwords per thread in registers, all
// loads issued before all stores, optional next-tile prefetch (PIPE) and LDS ballast to pin workgroups/CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/tilebench tools/tilebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned u32;

struct Geo {
    int rowdw_log;      // log2 dwords per row
    size_t stride;      // dwords between rows
    size_t chunk_step;  // dwords between consecutive column chunks (normally rowdw)
    int nchunks_log;    // log2 column chunks per frame
    size_t row_group;   // rows are numbered r = rlo + (rhi << rsplit): address = rlo * stride + rhi * row_group
    int rsplit;
};

// element e = j * T + tid of a tile: (thread base) + (uniform offset of register j)
__device__ __forceinline__ size_t thread_base(const Geo &g, unsigned tid)
{
    const unsigned col = tid & ((1u << g.rowdw_log) - 1u), row = tid >> g.rowdw_log;
    return (size_t)row * g.stride + col;
}
template <int T> __device__ __forceinline__ size_t reg_offset(const Geo &g, int j)
{
    const unsigned e = (unsigned)j * T;
    return (size_t)(e >> g.rowdw_log) * g.stride + (e & ((1u << g.rowdw_log) - 1u));
}

template <int T, int NREG, int NT, int PIPE>
__global__ __launch_bounds__(T) void tilecopy(const u32 *in, u32 *out, Geo gr, Geo gw, size_t frame_dw, unsigned nframes,
                                              int order)
{
    extern __shared__ u32 ballast[];
    const unsigned tid = threadIdx.x;
    const unsigned nch = 1u << gr.nchunks_log;
    const size_t ntiles = (size_t)nframes * nch;
    const size_t rbase = thread_base(gr, tid), wbase = thread_base(gw, tid);
    u32 v[NREG], w[NREG];
    auto where = [&](size_t t, size_t &frame, unsigned &chunk) {
        if (order == 2 || order == 3) { // the XCD pairing of k_big2x_a: blocks b and b + 8 (one XCD, same time) take the two halves of the same lines
            const unsigned slot = (unsigned)t & 7u, part = ((unsigned)t >> 3) & 1u;
            const unsigned half = nch >> 1; // column lines per frame
            if (order == 2) { // XCD s: lines s, s + 8, ... of every frame (shipped)
                const size_t G = (t >> 4) * 8u + slot;
                chunk = (unsigned)(G % half) * 2u + part;
                frame = G / half;
            } else { // XCD s: whole frames s, s + 8, ...
                const size_t jm = t >> 4;
                chunk = (unsigned)(jm % half) * 2u + part;
                frame = (jm / half) * 8u + slot;
            }
            if (frame >= nframes) frame = nframes - 1; // (ragged tails do not occur with nframes % 8 == 0)
        } else if (order == 4 || order == 5) { // quarter-line pieces (32 B): the four pieces of a line from blocks b, b + 8, b + 16, b + 24 of one XCD at
            // the same time (4), or from ONE block in four consecutive iterations (5)
            const unsigned lines = nch >> 2;
            size_t G;
            unsigned quad;
            if (order == 4) {
                const unsigned slot = (unsigned)t & 7u;
                quad = ((unsigned)t >> 3) & 3u;
                G = (t >> 5) * 8u + slot;
            } else {
                const size_t k = t / gridDim.x, b = t % gridDim.x;
                quad = (unsigned)k & 3u;
                G = (k >> 2) * gridDim.x + b;
            }
            chunk = (unsigned)(G % lines) * 4u + quad;
            frame = G / lines;
            if (frame >= nframes) frame = nframes - 1;
        } else if (order) {
            frame = t % nframes;
            chunk = (unsigned)(t / nframes);
        } else {
            frame = t / nch;
            chunk = (unsigned)(t % nch);
        }
    };
    size_t t = blockIdx.x;
    if (t >= ntiles) return;
#define TB_LOAD(TT, R)                                                                             \
    do {                                                                                           \
        size_t frame_;                                                                             \
        unsigned chunk_;                                                                           \
        where(TT, frame_, chunk_);                                                                 \
        const u32 *p_ = in + frame_ * frame_dw + (size_t)chunk_ * gr.chunk_step + rbase;           \
        _Pragma("unroll") for (int j = 0; j < NREG; ++j)                                           \
        {                                                                                          \
            const u32 *q_ = p_ + reg_offset<T>(gr, j);                                             \
            R[j] = NT ? __builtin_nontemporal_load(q_) : *q_;                                      \
        }                                                                                          \
    } while (0)
#define TB_STORE(TT, R)                                                                            \
    do {                                                                                           \
        size_t frame_;                                                                             \
        unsigned chunk_;                                                                           \
        where(TT, frame_, chunk_);                                                                 \
        u32 *p_ = out + frame_ * frame_dw + (size_t)chunk_ * gw.chunk_step + wbase;                \
        _Pragma("unroll") for (int j = 0; j < NREG; ++j)                                           \
        {                                                                                          \
            u32 *q_ = p_ + reg_offset<T>(gw, j);                                                   \
            if (NT) __builtin_nontemporal_store(R[j] + 1u, q_);                                    \
            else *q_ = R[j] + 1u;                                                                  \
        }                                                                                          \
    } while (0)
    if (PIPE) {
        TB_LOAD(t, v);
        for (; t < ntiles; t += gridDim.x) {
            const size_t tn = t + gridDim.x;
            if (tn < ntiles) TB_LOAD(tn, w);
            if (tid == 0xFFFFFFu) ballast[0] = v[0];
            __syncthreads();
            TB_STORE(t, v);
#pragma unroll
            for (int j = 0; j < NREG; ++j) v[j] = w[j];
        }
    } else {
        for (; t < ntiles; t += gridDim.x) {
            TB_LOAD(t, v);
            if (tid == 0xFFFFFFu) ballast[0] = v[0];
            __syncthreads();
            TB_STORE(t, v);
        }
    }
}

template <typename F> float timeit(F f, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

// rows x rowdw tile with column chunks side by side; rows at `stride`
static Geo strided(int rows_log, int rowdw_log, size_t stride, int frame_log)
{
    (void)rows_log;
    Geo g;
    g.rowdw_log = rowdw_log;
    g.stride = stride;
    g.chunk_step = (size_t)1 << rowdw_log;
    int sl = 0;
    while (((size_t)1 << sl) < stride) ++sl;
    g.nchunks_log = sl - rowdw_log;
    g.row_group = 0;
    g.rsplit = 30;
    (void)frame_log;
    return g;
}

int main(int argc, char **argv)
{
    const int frame_log = 20;
    const size_t frame_dw = (size_t)1 << frame_log;
    unsigned nframes = argc > 1 ? atoi(argv[1]) : 64;
    int iters = argc > 2 ? atoi(argv[2]) : 20;
    size_t bytes = (size_t)nframes * frame_dw * 4;
    void *in, *out;
    hipMalloc(&in, bytes);
    hipMalloc(&out, bytes);
    hipMemset(in, 1, bytes);
    hipMemset(out, 0, bytes);
    int cus;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("# frames=%u (%.0f MiB in + same out), CUs=%d; GB/s = read + write\n", nframes, bytes / 1048576.0, cus);

    auto rep = [&](const char *name, int wg_per_cu, float ms) {
        printf("%-64s wg/cu=%d  %8.3f ms  %7.1f GB/s\n", name, wg_per_cu, ms, 2.0 * bytes / ms / 1e6);
        fflush(stdout);
    };
#define RUN(NAME, T, NREG, NT, PIPE, GR, GW, LDSB, WGCU, ORDER)                                                                   \
    do {                                                                                                                          \
        hipFuncSetAttribute((const void *)tilecopy<T, NREG, NT, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);          \
        Geo gr_ = GR, gw_ = GW;                                                                                                    \
        unsigned grid_ = cus * WGCU;                                                                                               \
        float ms_ = timeit([&] { hipLaunchKernelGGL((tilecopy<T, NREG, NT, PIPE>), dim3(grid_), dim3(T), LDSB, 0, (const u32 *)in, \
                                                    (u32 *)out, gr_, gw_, frame_dw, nframes, ORDER); },                         \
                           iters);                                                                                                \
        char nm_[160];                                                                                                            \
        snprintf(nm_, sizeof nm_, "%s T=%d NREG=%d nt=%d pipe=%d order=%d", NAME, T, NREG, NT, PIPE, ORDER);                       \
        rep(nm_, WGCU, ms_);                                                                                                       \
    } while (0)

    // --- reference: today's pass 1 (256 rows x 128 B at 16 KiB stride; 512 threads x 16 regs; 2 WG/CU) ---
    {
        Geo g = strided(8, 5, 4096, frame_log);
        RUN("p1-today 256r x 128B @16K", 512, 16, 0, 0, g, g, 40 * 1024, 2, 1);
        RUN("p1-today 256r x 128B @16K", 512, 16, 1, 0, g, g, 40 * 1024, 2, 1);
        RUN("p1-today 256r x 128B @16K", 512, 16, 1, 0, g, g, 40 * 1024, 2, 0);
    }
    // --- candidate pass 1: 1024 rows at 4 KiB stride ---
    for (int order = 0; order < 2; ++order) {
        Geo g128 = strided(10, 5, 1024, frame_log), g64 = strided(10, 4, 1024, frame_log), g256 = strided(10, 6, 1024, frame_log);
        RUN("p1 1024r x 128B @4K", 1024, 32, 1, 0, g128, g128, 128 * 1024, 1, order);
        RUN("p1 1024r x 128B @4K", 1024, 32, 1, 1, g128, g128, 128 * 1024, 1, order);
        RUN("p1 1024r x  64B @4K", 1024, 16, 1, 0, g64, g64, 64 * 1024, 2, order);
        RUN("p1 1024r x  64B @4K", 512, 32, 1, 0, g64, g64, 64 * 1024, 2, order);
        RUN("p1 1024r x  64B @4K", 512, 32, 1, 1, g64, g64, 64 * 1024, 2, order);
        RUN("p1 1024r x 256B @4K", 1024, 64, 1, 0, g256, g256, 128 * 1024, 1, order);
    }
    // --- round 5: (a) the half-line tile of k_big2x_a WITH its XCD pairing (orders 2 / 3: blocks b, b + 8 take the two halves of the same lines),
    //     (b) 512 rows x 128 B at an 8 KiB stride (N = 2^9 x 2^11: full lines, 64 KiB, two workgroups per CU), (c) the same at 1 / 2 / 4 KiB strides
    for (int order = 2; order < 4; ++order) {
        Geo g64 = strided(10, 4, 1024, frame_log);
        RUN("p1 1024r x  64B @4K PAIRED", 512, 32, 1, 0, g64, g64, 68 * 1024, 2, order);
        RUN("p1 1024r x  64B @4K PAIRED plain loads", 512, 32, 0, 0, g64, g64, 68 * 1024, 2, order);
    }
    for (int order = 0; order < 2; ++order) {
        Geo g8k = strided(9, 5, 2048, frame_log);
        RUN("p1 512r x 128B @8K", 512, 32, 1, 0, g8k, g8k, 68 * 1024, 2, order);
        RUN("p1 512r x 128B @8K", 512, 32, 1, 1, g8k, g8k, 68 * 1024, 2, order);
        // write side of such a split: a 128 KiB tile (16 rows of 2048 points; the scratch layout is free: read as one contiguous run), written as
        // 2048 pieces of 64 B at a 2 KiB stride (X[k_low + 512 k_high]); one workgroup per CU
        Geo grb;
        grb.rowdw_log = 15, grb.stride = (size_t)1 << 15, grb.chunk_step = (size_t)1 << 15, grb.nchunks_log = 5, grb.row_group = 0, grb.rsplit = 30;
        Geo gwb = strided(11, 4, 512, frame_log);
        RUN("p2 read 128 KiB contiguous, write 2048 x 64B @2K", 1024, 32, 1, 0, grb, gwb, 132 * 1024, 1, order);
    }
    // the same split with a 64 KiB row tile (8 rows of 2048 points, two workgroups per CU): 2048 pieces of 32 B at a 2 KiB stride
    {
        Geo gr8;
        gr8.rowdw_log = 14, gr8.stride = (size_t)1 << 14, gr8.chunk_step = (size_t)1 << 14, gr8.nchunks_log = 6, gr8.row_group = 0, gr8.rsplit = 30;
        Geo gw8 = strided(11, 3, 512, frame_log);
        RUN("p2 read 64 KiB contiguous, write 2048 x 32B @2K", 512, 32, 1, 0, gr8, gw8, 68 * 1024, 2, 0);
        RUN("p2 read 64 KiB contiguous, write 2048 x 32B @2K XCD quads", 512, 32, 1, 0, gr8, gw8, 68 * 1024, 2, 4);
        RUN("p2 read 64 KiB contiguous, write 2048 x 32B @2K XCD quads, plain", 512, 32, 0, 0, gr8, gw8, 68 * 1024, 2, 4);
        RUN("p2 read 64 KiB contiguous, write 2048 x 32B @2K one block x 4", 512, 32, 1, 0, gr8, gw8, 68 * 1024, 2, 5);
        RUN("p2 read 64 KiB contiguous, write 2048 x 32B @2K one block x 4, plain", 512, 32, 0, 0, gr8, gw8, 68 * 1024, 2, 5);
    }
    // --- candidate pass 2: read 32 rows (n19..15) x 4 KiB contiguous, write 1024 runs of 128 B at 4 KiB stride ---
    //     read geometry: row = 1024 dwords, 32 rows at stride 2^15 dwords, 32 column chunks of 1024 dwords
    for (int order = 0; order < 2; ++order) {
        Geo gr;
        gr.rowdw_log = 10, gr.stride = (size_t)1 << 15, gr.chunk_step = 1024, gr.nchunks_log = 5, gr.row_group = 0, gr.rsplit = 30;
        Geo gw = strided(10, 5, 1024, frame_log);
        RUN("p2 read 32r x 4KiB, write 1024 x 128B @4K", 1024, 32, 1, 0, gr, gw, 128 * 1024, 1, order);
        RUN("p2 read 32r x 4KiB, write 1024 x 128B @4K", 1024, 32, 1, 1, gr, gw, 128 * 1024, 1, order);
        // 64-B variant: 16 rows x 4 KiB, write 1024 runs of 64 B
        Geo gr16 = gr;
        Geo gw64 = strided(10, 4, 1024, frame_log);
        // 16 rows: chunks = 64 per frame (2 per 32-row group): model as 64 chunks of 1024 dwords at rows stride 2^16
        gr16.stride = (size_t)1 << 16, gr16.nchunks_log = 6;
        RUN("p2 read 16r x 4KiB, write 1024 x  64B @4K", 512, 32, 1, 0, gr16, gw64, 64 * 1024, 2, order);
        RUN("p2 read 16r x 4KiB, write 1024 x  64B @4K", 512, 32, 1, 1, gr16, gw64, 64 * 1024, 2, order);
    }
    // --- contiguous in / contiguous out with the same machinery (ceiling of the harness) ---
    {
        auto contig = [](int tile_log) {
            Geo gc;
            gc.rowdw_log = tile_log, gc.stride = (size_t)1 << tile_log, gc.chunk_step = (size_t)1 << tile_log;
            gc.nchunks_log = 20 - tile_log, gc.row_group = 0, gc.rsplit = 30;
            return gc;
        };
        RUN("contiguous tiles", 1024, 32, 1, 0, contig(15), contig(15), 128 * 1024, 1, 0);
        RUN("contiguous tiles", 1024, 32, 1, 1, contig(15), contig(15), 128 * 1024, 1, 0);
        RUN("contiguous tiles", 512, 16, 1, 0, contig(13), contig(13), 40 * 1024, 2, 0);
        RUN("contiguous tiles", 256, 16, 1, 0, contig(12), contig(12), 20 * 1024, 4, 0);
    }
    return 0;
}
