// xcdbench.hip -- can the inter-pass data of a multi-pass transform stay on chip?  (diagnostic tool, not product)
//
// Part 1 (tiers): copy / read / write rates against the working-set size (L2 4 MiB per XCD, Infinity Cache 256 MiB, HBM).
// Part 2 (pipe):  ONE persistent launch that takes every 4 MiB frame through 2 or 3 tile passes (strided read ->
//                 scratch -> [in-place pass ->] strided write), frame-granular dependency counters instead of kernel
//                 boundaries, the scratch a small ring of frame slots.  Two placements:
//                   xcd = 1: a frame is worked on by the workgroups of ONE XCD (queue = HW_REG_XCC_ID), hand-off through
//                            that XCD's L2 (stores drained with vmcnt(0), consumer loads bypass L1), ring of S slots per XCD;
//                   xcd = 0: one queue for the chip, agent-scope release / acquire fences, ring of S slots in all.
//                 Every word of the first and last frame and a checksum of all frames are verified on the host.
//                 The baseline is the same tile passes as separate launches over the whole batch.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/xcdbench tools/xcdbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned u32;
typedef unsigned long long u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));            \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

template <typename F> static float timeit(F f, int warm, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < warm; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

// ---------------------------------------------------------------- part 1: tiers
template <int MODE, int NT> // MODE 0 copy, 1 read, 2 write
__global__ __launch_bounds__(256) void k_tier(const v4u *in, v4u *out, size_t n16, u32 *sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    v4u acc = {0, 0, 0, 0};
    for (; i < n16; i += st) {
        if (MODE == 2) {
            v4u v = {(u32)i, 1, 2, 3};
            if (NT) __builtin_nontemporal_store(v, out + i);
            else out[i] = v;
        } else {
            v4u v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            if (MODE == 0) {
                if (NT) __builtin_nontemporal_store(v, out + i);
                else out[i] = v;
            } else
                acc += v;
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345677u) sink[0] = 1;
}

static void tiers(int cus)
{
    const size_t maxb = (size_t)1 << 30;
    void *a, *b;
    u32 *sink;
    CK(hipMalloc(&a, maxb));
    CK(hipMalloc(&b, maxb));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, maxb));
    CK(hipMemset(b, 2, maxb));
    printf("# tiers: working set per array; GB/s (copy = read + write); grid = 8 blocks/CU x 256 threads, 16 B per lane\n");
    printf("%10s %10s %10s %10s %10s %10s %10s\n", "MiB/array", "copy", "copy_nt", "read", "read_nt", "write", "write_nt");
    for (size_t mb : {2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 512, 1024}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        const int iters = (int)((size_t)(16ull << 30) / bytes > 400 ? 400 : (16ull << 30) / bytes);
        float r[6];
        int k = 0;
#define T1(MODE, NT)                                                                                                       \
    r[k++] = timeit([&] { hipLaunchKernelGGL((k_tier<MODE, NT>), dim3(cus * 8), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, \
                                             n16, sink); },                                                                \
                    5, iters < 10 ? 10 : iters)
        T1(0, 0);
        T1(0, 1);
        T1(1, 0);
        T1(1, 1);
        T1(2, 0);
        T1(2, 1);
        printf("%10zu %10.0f %10.0f %10.0f %10.0f %10.0f %10.0f\n", mb, 2.0 * bytes / r[0] / 1e6, 2.0 * bytes / r[1] / 1e6,
               bytes / r[2] / 1e6, bytes / r[3] / 1e6, bytes / r[4] / 1e6, bytes / r[5] / 1e6);
        fflush(stdout);
    }
    hipFree(a);
    hipFree(b);
    hipFree(sink);
}

// ---------------------------------------------------------------- part 2: tile passes
struct Geo {            // a tile = T * NREG dwords: element e -> (e >> rowdw_log) * stride + (e & (rowdw - 1)), tile t at t_off(t)
    int rowdw_log;      // log2 dwords per row
    u32 stride;         // dwords between rows
    int nchunk_log;     // tiles per frame = 2^nchunk_log; tile c of a frame starts at (c & cmask) * cstep + (c >> clog) * cstep2
    u32 cstep, cstep2;
    int clog;
};
__device__ __host__ inline size_t tile_off(const Geo &g, u32 c)
{
    return (size_t)(c & ((1u << g.clog) - 1u)) * g.cstep + (size_t)(c >> g.clog) * g.cstep2;
}
__device__ __host__ inline size_t elem_off(const Geo &g, u32 e)
{
    return (size_t)(e >> g.rowdw_log) * g.stride + (e & ((1u << g.rowdw_log) - 1u));
}
static Geo strided_tile(int rowdw_log, u32 stride) // rows at `stride`, column chunks side by side
{
    Geo g;
    g.rowdw_log = rowdw_log, g.stride = stride;
    int sl = 0;
    while ((1u << sl) < stride) ++sl;
    g.nchunk_log = sl - rowdw_log, g.cstep = 1u << rowdw_log, g.cstep2 = 0, g.clog = 30;
    return g;
}
static Geo contig_tile(int tile_log)
{
    Geo g;
    g.rowdw_log = tile_log, g.stride = 1u << tile_log, g.nchunk_log = 20 - tile_log, g.cstep = 1u << tile_log, g.cstep2 = 0, g.clog = 30;
    return g;
}

struct Phase {
    Geo gr, gw;
    int src, dst; // 0 user in, 1 scratch, 2 user out
};
struct Pipe {
    Phase ph[3];
    int nph;
    int tiles_log;   // tiles per frame (same for every phase)
    u32 nframes;     // frames in all
    int slots;       // scratch ring slots per queue
    int lag;         // software-pipeline distance between consecutive phases, in frames
    int xcd;         // 1: queue = XCC id, L2 hand-off; 0: one queue, agent fences
    int nq;
};

template <int T, int NREG> __device__ __forceinline__ void copy_tile(const u32 *src, u32 *dst, const Geo &gr, const Geo &gw, bool src_scratch)
{
    const u32 tid = threadIdx.x;
    u32 v[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const u32 *q = src + elem_off(gr, (u32)j * T + tid);
        v[j] = __builtin_nontemporal_load(q); // nt: served by L2, never by this CU's L1
    }
    (void)src_scratch;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        u32 *q = dst + elem_off(gw, (u32)j * T + tid);
        *q = v[j] + 1u;
    }
}

// separate launches: one phase over the whole batch (scratch = a full-size array)
template <int T, int NREG>
__global__ __launch_bounds__(T) void k_phase(const u32 *src, u32 *dst, Geo gr, Geo gw, u32 nframes, int tiles_log)
{
    const size_t ntiles = (size_t)nframes << tiles_log;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t f = t >> tiles_log;
        const u32 c = (u32)t & ((1u << tiles_log) - 1u);
        copy_tile<T, NREG>(src + (f << 20) + tile_off(gr, c), dst + (f << 20) + tile_off(gw, c), gr, gw, false);
    }
}

__device__ __forceinline__ u32 xcc_id()
{
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; // HW_REG_XCC_ID[3:0]
}

// ctl layout (u32): [q][0] = ticket; [q][16 + phase * maxf + i] = done counter of (phase, frame sequence number i)
template <int T, int NREG>
__global__ __launch_bounds__(T) void k_pipe(const u32 *in, u32 *scratch, u32 *out, Pipe P, u32 *ctl, u32 ctl_stride, u32 maxf, u32 *err)
{
    __shared__ u32 s_ticket;
    const u32 q = P.xcd ? (xcc_id() % (u32)P.nq) : 0u;
    u32 *my = ctl + (size_t)q * ctl_stride;
    // frames of this queue: f = q + nq * i, i = 0 .. nf - 1
    const u32 nf = (P.nframes + (u32)P.nq - 1u - q) / (u32)P.nq;
    const u32 tiles = 1u << P.tiles_log;
    // rounds k = 0 .. nf - 1 + lag * (nph - 1); round k holds phase p of frame i = k - p * lag
    const u32 nrounds = nf + (u32)P.lag * (u32)(P.nph - 1);
    const u64 total = (u64)nrounds * (u32)P.nph * tiles;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(&my[0], 1u);
        __syncthreads();
        const u32 tk = s_ticket;
        __syncthreads();
        if ((u64)tk >= total) return;
        const u32 c = tk & (tiles - 1u);
        const u32 pr = tk >> P.tiles_log;
        const u32 p = pr % (u32)P.nph, k = pr / (u32)P.nph;
        const int i = (int)k - (int)p * P.lag;
        if (i < 0 || i >= (int)nf) continue;
        // dependencies
        if (threadIdx.x == 0) {
            const u32 *w = nullptr;
            if (p > 0) w = &my[16 + (p - 1) * maxf + i];
            else if (i >= P.slots) w = &my[16 + (u32)(P.nph - 1) * maxf + (i - P.slots)];
            if (w) {
                u32 spins = 0;
                while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tiles) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 24)) {
                        err[0] = 1;
                        break;
                    }
                }
            }
            if (!P.xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const Phase &ph = P.ph[p];
        const size_t f = (size_t)q + (size_t)P.nq * (u32)i;
        const size_t slot = ((size_t)q * P.slots + ((u32)i % (u32)P.slots)) << 20;
        const u32 *src = ph.src == 0 ? in + (f << 20) : scratch + slot;
        u32 *dst = ph.dst == 2 ? out + (f << 20) : scratch + slot;
        copy_tile<T, NREG>(src + tile_off(ph.gr, c), dst + tile_off(ph.gw, c), ph.gr, ph.gw, ph.src == 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (!P.xcd) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(&my[16 + p * maxf + i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static void host_phase(const std::vector<u32> &src, std::vector<u32> &dst, const Geo &gr, const Geo &gw, int tiles_log, u32 tile_elems)
{
    for (u32 c = 0; c < (1u << tiles_log); ++c)
        for (u32 e = 0; e < tile_elems; ++e) dst[tile_off(gw, c) + elem_off(gw, e)] = src[tile_off(gr, c) + elem_off(gr, e)] + 1u;
}

template <int T, int NREG> static void pipes(int cus, u32 nframes, int wgcu, int ldsb)
{
    const size_t fdw = (size_t)1 << 20;
    const size_t bytes = (size_t)nframes * fdw * 4;
    u32 *in, *out, *scr_full, *scr_ring, *ctl, *err;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMalloc(&scr_full, bytes));
    CK(hipMalloc(&scr_ring, (size_t)256 << 20));
    const u32 maxf = nframes, ctl_stride = 16 + 3 * maxf;
    CK(hipMalloc(&ctl, (size_t)8 * ctl_stride * 4));
    CK(hipMalloc(&err, 64));
    CK(hipMemset(err, 0, 64));
    std::vector<u32> hin((size_t)nframes * fdw);
    u64 sum_in = 0;
    for (size_t i = 0; i < hin.size(); ++i) {
        hin[i] = (u32)(i * 2654435761u) >> 4;
        sum_in += hin[i];
    }
    CK(hipMemcpy(in, hin.data(), bytes, hipMemcpyHostToDevice));
    const int tile_log = 0;
    (void)tile_log;
    const u32 tile_elems = T * NREG;
    int tl = 0;
    while ((1u << tl) < tile_elems) ++tl;
    const int tiles_log = 20 - tl;
    // the tile shapes: strided = 2^(tl-5) rows of 128 B; contiguous = the tile in one run
    const Geo gs = strided_tile(5, 1u << (20 - (tl - 5))), gc = contig_tile(tl);
    printf("# pipes: %u frames of 4 MiB (%.0f MiB in + same out), tile = %u dwords (%d per frame), T=%d NREG=%d, %d WG/CU\n", nframes,
           bytes / 1048576.0, tile_elems, 1 << tiles_log, T, NREG, wgcu);
    std::vector<u32> hout((size_t)nframes * fdw);
    auto verify = [&](const char *name, int nph, const Phase *ph) -> bool {
        CK(hipMemcpy(hout.data(), out, bytes, hipMemcpyDeviceToHost));
        u32 herr;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        u64 s = 0;
        for (size_t i = 0; i < hout.size(); ++i) s += hout[i];
        bool ok = (s == sum_in + (u64)nph * hout.size()) && !herr;
        for (u32 f : {0u, nframes - 1}) {
            std::vector<u32> a(hin.begin() + f * fdw, hin.begin() + (f + 1) * fdw), b(fdw);
            for (int p = 0; p < nph; ++p) {
                host_phase(a, b, ph[p].gr, ph[p].gw, tiles_log, tile_elems);
                a.swap(b);
            }
            ok = ok && memcmp(a.data(), hout.data() + f * fdw, fdw * 4) == 0;
        }
        if (!ok) printf("!! %s: WRONG RESULT (err flag %u)\n", name, herr);
        return ok;
    };
    hipFuncSetAttribute((const void *)k_phase<T, NREG>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipFuncSetAttribute((const void *)k_pipe<T, NREG>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    const u32 grid = cus * wgcu;
    for (int nph = 2; nph <= 3; ++nph) {
        Phase ph[3];
        // 2 phases: strided read -> contiguous scratch tile; contiguous scratch tile -> strided write
        // 3 phases: the same with an in-place contiguous pass on the scratch in the middle
        ph[0] = {gs, gc, 0, 1};
        if (nph == 3) ph[1] = {gc, gc, 1, 1};
        ph[nph - 1] = {gc, gs, 1, 2};
        // baseline: separate launches over the whole batch
        CK(hipMemset(out, 0, bytes));
        auto base = [&] {
            hipLaunchKernelGGL((k_phase<T, NREG>), dim3(grid), dim3(T), ldsb, 0, in, scr_full, ph[0].gr, ph[0].gw, nframes, tiles_log);
            if (nph == 3)
                hipLaunchKernelGGL((k_phase<T, NREG>), dim3(grid), dim3(T), ldsb, 0, scr_full, scr_full, ph[1].gr, ph[1].gw, nframes, tiles_log);
            hipLaunchKernelGGL((k_phase<T, NREG>), dim3(grid), dim3(T), ldsb, 0, scr_full, out, ph[nph - 1].gr, ph[nph - 1].gw, nframes,
                               tiles_log);
        };
        float ms = timeit(base, 3, 10);
        bool ok = verify("separate", nph, ph);
        printf("%d passes, separate launches, full-size scratch                      %8.3f ms  %7.1f Gsample/s  %s\n", nph, ms,
               nframes * 1048576.0 / ms / 1e6, ok ? "ok" : "WRONG");
        fflush(stdout);
        for (int xcd = 1; xcd >= 0; --xcd)
            for (int lag = 0; lag <= 2; ++lag)
                for (int slots : {1, 2, 3, 4, 8}) {
                    if (slots < 1 + lag * (nph - 1) && !(lag == 0 && slots == 1)) continue;
                    if (lag == 0 && slots > 2) continue;
                    if (xcd == 0 && slots * 4 > 256) continue;
                    Pipe P;
                    memset(&P, 0, sizeof P);
                    for (int p = 0; p < nph; ++p) P.ph[p] = ph[p];
                    P.nph = nph, P.tiles_log = tiles_log, P.nframes = nframes, P.slots = xcd ? slots : slots * 8, P.lag = xcd ? lag : lag * 8;
                    P.xcd = xcd, P.nq = xcd ? 8 : 1;
                    CK(hipMemset(out, 0, bytes));
                    auto run = [&] {
                        hipMemsetAsync(ctl, 0, (size_t)8 * ctl_stride * 4, 0);
                        hipLaunchKernelGGL((k_pipe<T, NREG>), dim3(grid), dim3(T), ldsb, 0, in, scr_ring, out, P, ctl, ctl_stride, maxf, err);
                    };
                    ms = timeit(run, 2, 10);
                    ok = verify("pipe", nph, ph);
                    printf("%d passes, ONE launch, %s, lag %d frames, %3d slots (%3d MiB ring)   %8.3f ms  %7.1f Gsample/s  %s\n", nph,
                           xcd ? "per-XCD queues (L2 hand-off)" : "one queue (agent fences)   ", P.lag, P.slots, P.slots * 4 * P.nq, ms,
                           nframes * 1048576.0 / ms / 1e6, ok ? "ok" : "WRONG");
                    fflush(stdout);
                }
    }
    hipFree(in), hipFree(out), hipFree(scr_full), hipFree(scr_ring), hipFree(ctl), hipFree(err);
}


// ---------------------------------------------------------------- part 3: narrow rows, the halves of a 128-byte line on ONE XCD
// A 10 + 10 plan for N = 2^20 wants 1024-row tiles; with 128-byte rows that is a 128 KiB tile = one workgroup per CU.  With
// 64-byte rows (two workgroups per CU) the pure copy drops to 2.5 TB/s when the two halves of a line are handled by unrelated
// workgroups (tilebench).  Here the two (or four) workgroups that share a line are blocks b and b + 8 (+ 16, + 24): same XCD
// (block b runs on XCD b % 8), same time -> the line is fetched into / written back from that L2 once.
//   map 0: tile t -> frame t / tiles, chunk t % tiles;   map 1: chunk = (column group, part) with part = (t >> 3) % PARTS
template <int T, int NREG>
__global__ __launch_bounds__(T) void k_phase_map(const u32 *src, u32 *dst, Geo gr, Geo gw, u32 nframes, int tiles_log, int parts_log, int map)
{
    const size_t ntiles = (size_t)nframes << tiles_log;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t f;
        u32 c;
        if (map) {
            const u32 parts = 1u << parts_log;
            const size_t blk = t / (8u * parts);             // a block of 8 * parts consecutive tiles: 8 column groups x parts
            const u32 u = (u32)(t % (8u * parts));
            const size_t G = blk * 8u + (u & 7u);            // column group number over the batch
            const u32 part = u >> 3;
            const u32 groups_per_frame = 1u << (tiles_log - parts_log);
            f = G / groups_per_frame;
            c = (u32)(G % groups_per_frame) * parts + part;
        } else {
            f = t >> tiles_log;
            c = (u32)t & ((1u << tiles_log) - 1u);
        }
        copy_tile<T, NREG>(src + (f << 20) + tile_off(gr, c), dst + (f << 20) + tile_off(gw, c), gr, gw, false);
    }
}

template <int T, int NREG> static void pair_case(int cus, u32 nframes, int wgcu, int ldsb, int rowdw_log, u32 *in, u32 *out)
{
    const size_t bytes = (size_t)nframes << 22;
    const u32 tile_elems = T * NREG;
    int tl = 0;
    while ((1u << tl) < tile_elems) ++tl;
    const int tiles_log = 20 - tl, parts_log = 5 - rowdw_log;
    const int rows_log = tl - rowdw_log;
    // strided side: 2^rows_log rows of 2^rowdw_log dwords at stride 2^(20 - rows_log) dwords; chunks side by side
    const Geo gs = strided_tile(rowdw_log, 1u << (20 - rows_log));
    // contiguous-row side (second pass reading whole 2^(20-rows_log)... rows): 2^(tl - 10) rows of 1024 dwords, rows at stride 2^(30 - tl)
    Geo gc;
    gc.rowdw_log = 10, gc.stride = 1u << (30 - tl), gc.nchunk_log = tiles_log, gc.cstep = 1024, gc.cstep2 = 0, gc.clog = 30;
    const Geo gt = contig_tile(tl);
    hipFuncSetAttribute((const void *)k_phase_map<T, NREG>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    const u32 grid = cus * wgcu;
    for (int shape = 0; shape < 3; ++shape)
        for (int map = 0; map <= (parts_log ? 1 : 0); ++map) {
            const Geo gr = shape == 0 ? gs : shape == 1 ? gs : gc;
            const Geo gw = shape == 0 ? gs : shape == 1 ? gt : gs;
            float ms = timeit([&] { hipLaunchKernelGGL((k_phase_map<T, NREG>), dim3(grid), dim3(T), ldsb, 0, in, out, gr, gw, nframes, tiles_log,
                                                       parts_log, map); },
                              3, 10);
            printf("%4d rows x %3d B, T=%4d NREG=%d, %d WG/CU, %-44s %-10s %8.3f ms  %7.1f GB/s\n", 1 << rows_log, 4 << rowdw_log, T, NREG, wgcu,
                   shape == 0 ? "strided read -> strided write (in place)" : shape == 1 ? "strided read -> contiguous tile write" : "4 KiB rows read -> strided write",
                   map ? "XCD-paired" : "unpaired", ms, 2.0 * bytes / ms / 1e6);
            fflush(stdout);
        }
}

static void pairs(int cus, u32 nframes)
{
    const size_t bytes = (size_t)nframes << 22;
    u32 *in, *out;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes));
    CK(hipMemset(out, 0, bytes));
    printf("# pairs: %u frames (%.0f MiB in + same out); GB/s = read + write\n", nframes, bytes / 1048576.0);
    pair_case<1024, 32>(cus, nframes, 1, 128 * 1024, 5, in, out); // 1024 rows x 128 B
    pair_case<512, 32>(cus, nframes, 2, 64 * 1024, 4, in, out);    // 1024 rows x 64 B
    pair_case<256, 32>(cus, nframes, 4, 36 * 1024, 3, in, out);    // 1024 rows x 32 B
    pair_case<512, 16>(cus, nframes, 2, 64 * 1024, 5, in, out);    // 256 rows x 128 B (today's pass 1)
    pair_case<256, 16>(cus, nframes, 4, 36 * 1024, 4, in, out);    // 256 rows x 64 B
    pair_case<512, 16>(cus, nframes, 4, 36 * 1024, 4, in, out);    // 512 rows x 64 B
    hipFree(in), hipFree(out);
}

// ---------------------------------------------------------------- part 4: load / store cache-policy flavours
// One wave per 4 KiB frame, 16 dword loads + 16 dword stores per lane (the headline kernel's pattern), 512 MiB in + 512 MiB out.
// LD: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1.   ST: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1
template <int LD> __device__ __forceinline__ u32 ld_flavour(const u32 *p)
{
    u32 v;
    if (LD == 0) return *p;
    if (LD == 1) return __builtin_nontemporal_load(p);
    if (LD == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (LD == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int ST> __device__ __forceinline__ void st_flavour(u32 *p, u32 v)
{
    if (ST == 0) *p = v;
    else if (ST == 1) __builtin_nontemporal_store(v, p);
    else if (ST == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (ST == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (ST == 4) asm volatile("global_store_dword %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off nt sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD, int ST> __global__ __launch_bounds__(256) void k_flavour(const u32 *in, u32 *out, size_t nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t f = (size_t)blockIdx.x * 4 + wv; f < nframes; f += (size_t)gridDim.x * 4) {
        const u32 *s = in + f * 1024 + lane;
        u32 *d = out + f * 1024 + lane;
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_flavour<LD>(s + 64 * j);
        if (LD >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) st_flavour<ST>(d + 64 * j, v[j] + 1u);
    }
}
static void flavours(int cus)
{
    const size_t nframes = 131072, bytes = nframes * 4096;
    u32 *in, *out;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes));
    CK(hipMemset(out, 0, bytes));
    printf("# flavours: %zu frames of 4 KiB (%.0f MiB in + same out), grid = 4 blocks/CU; GB/s = read + write\n", nframes, bytes / 1048576.0);
    const char *ldn[4] = {"plain", "nt", "sc1", "sc0 sc1"}, *stn[6] = {"plain", "nt", "sc1", "sc0 sc1", "nt sc1", "nt sc0 sc1"};
#define FL(LD, ST)                                                                                                                           \
    {                                                                                                                                       \
        float ms = timeit([&] { hipLaunchKernelGGL((k_flavour<LD, ST>), dim3(cus * 4), dim3(256), 0, 0, in, out, nframes); }, 30, 50);        \
        printf("load %-8s store %-11s %8.3f ms  %7.1f GB/s\n", ldn[LD], stn[ST], ms, 2.0 * bytes / ms / 1e6);                               \
        fflush(stdout);                                                                                                                     \
    }
    FL(0, 0) FL(0, 1) FL(0, 2) FL(0, 3) FL(0, 4) FL(0, 5)
    FL(1, 0) FL(1, 1) FL(1, 4)
    FL(2, 1) FL(3, 1)
    hipFree(in), hipFree(out);
}

int main(int argc, char **argv)
{
    int cus;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const char *what = argc > 1 ? argv[1] : "all";
    u32 nframes = argc > 2 ? atoi(argv[2]) : 64;
    if (!strcmp(what, "tiers") || !strcmp(what, "all")) tiers(cus);
    if (!strcmp(what, "pipes") || !strcmp(what, "all")) {
        pipes<512, 16>(cus, nframes, 2, 40 * 1024);
        pipes<1024, 32>(cus, nframes, 1, 128 * 1024);
    }
    if (!strcmp(what, "pairs") || !strcmp(what, "all")) pairs(cus, nframes);
    if (!strcmp(what, "flavours") || !strcmp(what, "all")) flavours(cus);
    return 0;
}
