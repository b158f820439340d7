// intfft_internal.hpp -- host<->kernel contracts inside libintfft.so (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "intfft_device.hpp"

#include <cstdlib>

namespace intfft {
// Diagnostic switches (A/B parity of kernel families in the tests, tuning experiments): the INTFFT_* variables listed in
// include/intfft.h are honoured ONLY when INTFFT_DIAG=1 is set as well -- a stray INTFFT_* variable in a production environment
// changes nothing.  Read per call (the tests flip them between plans).
inline const char *diag_env(const char *name)
{
    const char *on = std::getenv("INTFFT_DIAG");
    return (on && on[0] == '1') ? std::getenv(name) : nullptr;
}
} // namespace intfft

namespace intfft {

constexpr int MAX_STAGES_PER_PASS = 40;
constexpr int PASS_THREADS = 256;

// Where a pass reads from / writes to.
enum IoMode : int {
    IO_SCRATCH = 0, // plan-owned scratch: words of the compute type at the core (in-place) index
    IO_USER = 1,    // ABI buffer: int16/32/64 containers in one of the INTFFT_ORDER_* layouts
};

// One kernel launch of the generic LDS pass kernel: a tile of 2^U points per frame is loaded into
// LDS, `nstages` radix-2 stages are evaluated in place, and the tile is stored.
//
// The core index j (0 <= j < 2^L) is the flat in-place index of SURVEY.md section 9.1: a DIF stage with
// STAGE = s pairs j and j + 2^s and uses twiddle j mod 2^s; after all DIF stages position j holds
// X[bitrev(j)]; the DIT stages run the mirror.  A tile owns U of the L index bits, given as two
// runs: tile-local bits [0, len0) sit at index bits [pos0, pos0+len0) and local bits [len0, U) at
// [pos1, pos1+len1); the block id supplies the remaining L-U bits in ascending order.
struct PassArgs {
    int L;        // log2 of the transform length
    int U;        // log2 of the tile length
    int len0, pos0, len1, pos1;
    int fpb;      // frames per block (power of two; > 1 only when U == L)
    int nstages;
    int in_mode, out_mode;   // IoMode
    int in_cb, out_cb;       // container bytes per component (user mode)
    int in_order, out_order; // INTFFT_ORDER_* (user mode)
    int in_rev, out_rev;     // user mode: logical index = bitrev(core index)  (frequency side)
    int in_bits;             // user mode: DATA_WIDTH the samples are wrapped to on load
    int in_zext;             // user mode: zero-extend instead (USE_FLY='0' unscaled)
    int ld_swap, st_swap;    // iterate the tile with the two runs swapped (coalescing of run1)
    int ld_memorder, st_memorder; // U == L only: sweep the frame in MEMORY order (coalesced global side,
                                  // the permutation is absorbed by the LDS side)
    int word;        // k_pass<T>: bytes of this pass's on-chip word (4 / 8); 0 = the plan's word
    int scr_in_word; // k_pass<T>: bytes of the scratch words this pass reads (<= word: an earlier, narrower pass)
    StageDesc st[MAX_STAGES_PER_PASS];
};

// ---- launch geometry (intfft_plan.hip) ------------------------------------------------------------------
// Persistent kernels launch exactly their resident grid.  resident_blocks() = CUs x blocks per CU on the calling
// thread's current device: the occupancy query for `threads`-wide blocks (`dflt` when it fails), clamped to
// `max_per_cu` when that is > 0, replaced by INTFFT_BLOCKS_PER_CU when `env_override`.  Results are cached per
// (kernel, device) under a mutex, so plans on different devices and concurrent callers each see their own entry.
size_t resident_blocks(const void *kernel, int threads, int dflt, int max_per_cu = 0, bool env_override = true);
int device_cus(); // CUs of the current device (cached per device)
// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KiB for `kernel`, once per (kernel, device)
void allow_max_lds(const void *kernel);
template <typename K> inline const void *kptr(K k) { return reinterpret_cast<const void *>(k); }

// generic kernels (intfft_generic.hip)
hipError_t launch_pass(const PassArgs &a, int word_bytes, const void *in, void *out, const int2 *tw,
                       size_t nframes, hipStream_t stream, const int2 *tw2d = nullptr);
size_t pass_lds_bytes(const PassArgs &a, int word_bytes);
unsigned pass_threads(const PassArgs &a);
const char *pass_kernel_name(int word_bytes);

hipError_t launch_twiddle_stage(const int2 *d_rom, int stage, int twd, int xser, int2 *d_out,
                                hipStream_t stream);

// packed int16 LDS pass kernel (intfft_pass16.hip): scaled, DATA_WIDTH = 16, TWDL_WIDTH <= 16
bool pass16_supported(int data_width, int twdl_width, int format, int use_fly);
hipError_t launch_pass16(const PassArgs &a, const void *in, void *out, const uint2 *twf, const uint2 *twi,
                         size_t nframes, int twd, hipStream_t stream);
hipError_t launch_pack_twiddles16(const int2 *tw, size_t n, uint2 *f, uint2 *i, hipStream_t stream);
const char *pass16_kernel_name();

// packed int16 lane-per-frame kernel for N = 8, 16, 32 (intfft_fastsmall.hip)
bool fastsmall_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly,
                         int in_order, int out_order);
hipError_t launch_fastsmall(int log2n, int direction, int rnd_round, int twd, const void *in, void *out, const int2 *h_tw,
                            size_t nframes, hipStream_t stream, int data_width = 16);
const char *fastsmall_kernel_name();

// ---- wave kernels (intfft_fast1024.hip, intfft_fast1024u.hip): I/O permutation for short frames ----
// Frames shorter than 1024 samples (L = log2 N in 6..9): the wave owns a chunk of 1024 consecutive samples
// = 2^(10-L) whole frames; chunk index bits a9..aL number the frame, a(L-1)..a0 the sample.  The stages of
// the frame-number bits are skipped (the twiddle index of STAGE s is the position mod 2^s, so the remaining
// stages are unchanged) and only the I/O permutation differs.  lane_bit<L>(k): the lane bit that carries
// index bit a_k (k = 4..9) after the LDS transpose.
//   L = 10: lane bit i = a(9-i), so that X index = rev4(r) * 64 + lane (256-byte runs per store).
//   L < 10: lane bits 5, 4 = a(L-1), a(L-2): two lane swaps after the last stage bring them into registers,
//           every lane then owns 4 consecutive outputs (brev_L puts a(L-1), a(L-2) into output bits 0, 1) and
//           stores them as one dwordx4; lane bits 0.. carry a(L-3), a(L-4).. (output bits 2, 3..), then the
//           frame bits.
template <int L> __host__ __device__ constexpr int lane_bit(int k)
{
    if (L == 10) return 9 - k;
    if (k == L - 1) return 5;
    if (k == L - 2) return 4;
    if (k < L - 2) return (L - 3) - k;
    return (L - 6) + (k - L);
}
// weight (dwords) of index bit a_k in the natural-order output of a chunk: in-frame bits are bit-reversed
template <int L> __host__ __device__ constexpr int out_weight(int k) { return k >= L ? (1 << k) : (1 << (L - 1 - k)); }

// packed int16 wave kernel for N = 1024 (intfft_fast1024.hip)
// DATA_WIDTH of the packed int16 kernels (intfft_fast1024*.hip, intfft_fast4096.hip): 16, or 9 .. 15 in truncate mode -- narrow data
// runs in the same 16-bit lanes (Slice::set_width, intfft_pk16.hpp); INTFFT_NO_NARROW16 sends it back to the 32-bit kernels
bool packed_width_ok(int data_width, int format, int rndmode);
struct Fast1024Args {
    int dw = 16;    // DATA_WIDTH (9 .. 16)
    int log2n;      // 6..10: frames shorter than 1024 samples share a wave (2^(10 - log2n) per pass)
    int twd;        // twiddle width (<= 16)
    int rnd;        // RoundKind (RND_TRUNC / RND_ROUND)
    int out_bitrev; // 0: NATURAL output, 1: BITREV output, 2: BITREV_LANES output (the BITREV kernels with the serial-stream store map)
    int in_halves;  // 0: NATURAL input, 1: HALVES input (native int_fftNk beats)
};
bool fast1024_supported(int log2n, int data_width, int twdl_width, int format, int rndmode,
                        int direction, int use_fly, int in_order, int out_order);
// tw_all: device twiddle buffer (stage s at offset 2^s - 1); h_tw: the host copy of the same
hipError_t launch_fast1024(const Fast1024Args &a, const void *in, void *out, const int2 *tw_all,
                           const int2 *h_tw, size_t nframes, hipStream_t stream);
const char *fast1024_kernel_name();

// packed int16 wave kernel for N = 1024, INV / PAIR (intfft_fast1024x.hip)
bool fast1024x_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction,
                         int use_fly, int in_order, int out_order);
hipError_t launch_fast1024x(int log2n, int direction, int twd, int in_bitrev, int out_halves, const void *in, void *out,
                            const int2 *tw_all, const int2 *h_tw,
                            size_t nframes, hipStream_t stream, int round = 0, int data_width = 16);
const char *fast1024x_kernel_name();

// unscaled int32 wave kernel for N = 1024, 16-bit in -> 26-bit out (intfft_fast1024u.hip)
bool fast1024u_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly,
                         int in_order, int out_order);
hipError_t launch_fast1024u(int log2n, int twd, const void *in, void *out, const int2 *tw_all, const int2 *h_tw, size_t nframes,
                            hipStream_t stream, int native = 0); // native: bit 0 HALVES in, bit 1 BITREV out
const char *fast1024u_kernel_name();

// unscaled int32 wave kernel, inverse core and pair (intfft_fast1024ux.hip)
struct UxStage {
    int sh;        // a + b of the multiplier regime
    unsigned keep; // ~(2^a - 1)
    int w;         // multiplier width (DTW of the DIT stage): T is wrapped to w bits
};
struct UxArgs {
    UxStage st[10]; // by STAGE number of the inverse core
};
bool fast1024ux_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly,
                          int in_order, int out_order);
hipError_t launch_fast1024ux(int log2n, int direction, int twd, const UxArgs &a, const void *in, void *out,
                             const int2 *tw_all, const int2 *h_tw, size_t nframes, hipStream_t stream, int native = 0); // native (inverse core): bit 0 HALVES out, bit 1 BITREV in
const char *fast1024ux_kernel_name();

// general-width int32 wave kernel, N = 64..1024 (intfft_fastw32.hip)
struct W32Stage {
    int sh;        // a + b of the multiplier regime
    unsigned keep; // ~(2^a - 1)
    int wsh;       // 32 - output width of the stage (multiplier output wrap)
    int wosh;      // 32 - output width (round-mode sum / difference wrap)
};
struct W32Args {
    W32Stage st[20]; // by STAGE number (16 .. 19: k_bigw_pre, intfft_bigwlong.hip)
    int in16, out16; // containers: 1 = int16 pairs, 0 = int32 pairs
    int in_sh;       // 32 - DATA_WIDTH
    int inverse;     // stage records describe int_ifftNk (DIT)
    int out64;       // 1: unscaled results of 33 / 34 bits: stages 1, 0 in 64 bits, int64 containers;
                     // 2 (N = 2048 / 4096): results of 35 / 36 bits: the whole last register round (STAGE 3..0) in 64 bits
    int masked;      // some stage is in a multi-DSP regime (a > 0): use the masked multiplier form
    int two_pass;    // N = 2^13 .. 2^16 forward: k_bigw_a + k_bigw_b instead of the three passes
    int native;      // k_bigw_a/b, qb/qa (round 5): bit 0 HALVES order on the time side, bit 1 BITREV order on the frequency side
};
bool fastw32_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                       int out_order);
hipError_t launch_fastw32(int log2n, int mode, const W32Args &a, const void *in, void *out, const int2 *tw_all,
                          const int2 *h_tw, size_t nframes, hipStream_t stream, int native = 0); // native: bit 0 HALVES in, bit 1 BITREV out
const char *fastw32_kernel_name();
// general-width int32 block kernel, N = 2048 / 4096 (intfft_fast4096w.hip)
bool fast4096w_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                         int out_order);
hipError_t launch_fast4096w(int log2n, int mode, const W32Args &a, const void *in, void *out, const int2 *tw_all,
                            const int2 *h_tw, size_t nframes, hipStream_t stream, int native = 0); // native: bit 0 HALVES in, bit 1 BITREV out
const char *fast4096w_kernel_name();
// general-width inverse kernels, N = 64..4096 (intfft_w32inv.hip)
bool w32inv_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                      int out_order);
hipError_t launch_w32inv(int log2n, int mode, const W32Args &a, const void *in, void *out, const int2 *tw_all,
                         const int2 *h_tw, size_t nframes, hipStream_t stream, int native = 0); // native: bit 0 HALVES out, bit 1 BITREV in
const char *w32inv_kernel_name(int log2n);
// general-width three-pass kernels, N = 2^13 .. 2^16 (intfft_bigw.hip)
bool bigw_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                    int out_order);
hipError_t launch_bigw(int log2n, int mode, const W32Args &a, const void *in, void *out, void *scratch, const int2 *tw_all,
                       const int2 *h_tw, size_t nframes, hipStream_t stream);
const char *bigw_kernel_name(int direction, int two_pass);
// the same class at N = 2^17 .. 2^20, forward: k_bigw_pre (STAGE NFFT-1 .. 16) + k_bigw_a<16> in place on the blocks + k_bigw_b (intfft_bigwlong.hip, round 5)
bool bigw_long_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order);
hipError_t launch_bigw_pre(int log2n, int mode, const W32Args &a, const void *in, int2 *scr, const int2 *tw, size_t nframes, hipStream_t stream);
hipError_t launch_bigw_post(int log2n, int mode, const W32Args &a, const int2 *scr, void *out, const int2 *tw, size_t nframes, hipStream_t stream);

// two-pass kernels for N = 65536, 24-bit unscaled, int32 in -> int64 out (intfft_wide16.hip)
struct WideStage {
    int sh;            // a + b: bit offset of the result slice in the 64-bit sum
    unsigned keep;     // ~(2^a - 1): per-product pre-truncation as a mask of the low dword
    int az;            // a == 0 (exact sum first)
    int s2, s3;        // 32-bit stages: alignbit amount sh + wo - 32, sign shift 32 - wo
    int w32;           // 64-bit stages: wo - 32
};
struct WideArgs {
    WideStage st[20];  // processing order: forward st[ii] is STAGE NFFT - 1 - ii, inverse st[s] is STAGE s (N = 2^17 .. 2^20: intfft_widelong.hip)
    int dw;            // DATA_WIDTH (wrap on load)
    int native;        // bit 0: HALVES order on the time side, bit 1: BITREV order on the frequency side (NAT instantiations)
    int r32;           // long frames: STAGE 7 .. 4 within 32 bits (k_wide16_p2<.., R32>)
    int w64;           // the first pass runs on 64-bit words too (k_wide64_p1 / q1: DATA_WIDTH 25 .. 32), 16-byte scratch samples
};
bool wide16_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                      int out_order);
hipError_t launch_wide16(int log2n, const WideArgs &a, const void *in, void *out, void *scratch, const int2 *tw_all,
                         const int2 *h_tw, size_t nframes, hipStream_t stream, int direction = 0);
const char *wide16_kernel_name(int direction = 0, int w64 = 0);
int wide16_class(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order);
// the same class at N = 2^17 .. 2^20: a pre-pass for STAGE NFFT-1 .. 16, then the two passes on 2^16-point blocks (intfft_widelong.hip, round 5)
int widelong_class(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order); // 0 none, 1 / 2 as wide16_class
hipError_t launch_widelong(int log2n, const WideArgs &a, int in_cb, const void *in, void *out, void *scratch, const int2 *tw_all, const int2 *h_tw,
                           size_t nframes, hipStream_t stream, int direction = 0);

// three-pass packed int16 kernels for N = 2^20 forward, natural -> natural (intfft_big20.hip)
bool big20_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly,
                     int in_order, int out_order);
hipError_t launch_big20(int log2n, int twd, int in_halves, int out_bitrev, int two_pass, const void *in, void *out, void *scratch,
                        const int2 *tw_all, const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream, int data_width = 16, int rndmode = 0);
const char *big20_kernel_name(int direction, int two_pass, int freq_bitrev);
hipError_t launch_bigpair(int log2n, int twd, int two_pass, const void *in, void *out, void *scratch, const int2 *tw_all,
                          const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream, int data_width = 16, int rndmode = 0);
hipError_t launch_biginv(int log2n, int twd, int in_bitrev, int out_halves, int two_pass, const void *in, void *out, void *scratch,
                         const int2 *tw_all, const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream, int data_width = 16, int rndmode = 0);
// two-pass plans for N = 2^19, 2^20 forward: 1024 x 1024 split on half lines, XCD-paired blocks (intfft_big2x.hip)
bool big2x_supported(int log2n);
bool big2x_tables_ok(int log2n, const int2 *h_tw, int twd);
const char *big2x_kernel_name();
hipError_t launch_big2x(int log2n, bool fx, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw16f, const int2 *h_tw, size_t nframes,
                        const struct Slice &sl, int halves, hipStream_t stream, bool out_bitrev = false);
// 64-bit wave kernels: N = 64 .. 1024 forward / inverse with results of 33 .. 64 bits (intfft_fastw64.hip)
bool fastw64_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order);
const char *fastw64_kernel_name(int direction);
bool fastw64_plan_ok(int log2n, const StageDesc *st10, int rnd_kind); // short frames: the narrow multiplier form only
// ... and N = 2048 / 4096 on the block kernels (intfft_fastw64b.hip, intfft_fastw64bi.hip)
bool fastw64b_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order);
bool fastw64b_plan_ok(int log2n, const StageDesc *st12, int rnd_kind);
const char *fastw64b_kernel_name(int direction);
hipError_t launch_fastw64b(int log2n, int direction, int rnd_kind, const StageDesc *st12, int in_cb, int dw, const void *in, void *out, const int2 *tw_all,
                           const int2 *h_tw, size_t nframes, hipStream_t stream, int native = 0); // native: bit 0 HALVES on the time side, bit 1 BITREV on the frequency side
hipError_t launch_fastw64(int log2n, int direction, int rnd_kind, const StageDesc *st10, int in_cb, int dw, const void *in, void *out, const int2 *tw_all, const int2 *h_tw,
                          size_t nframes, hipStream_t stream, int native = 0); // native (N = 1024): bit 0 HALVES on the time side, bit 1 BITREV on the frequency side
// the 2-D scheme at N = 2^20 = 1024 x 1024 in two launches (intfft_big2x.hip): column cores + multiplier, row cores + store
int fused2d_supported(int log2n, int l1, int data_width, int twdl_width, int format, int rndmode, int direction, int in_order, int out_order);
hipError_t build_fused2d_table(uint32_t *d_table, int log2n, int twd, hipStream_t stream, int l1 = 10);
hipError_t launch_fused2d_2k2k(int twd, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw16r, const int2 *h_tw, const uint32_t *tw2d, size_t nframes,
                               hipStream_t stream); // N = 2^22 = 2048 x 2048 in two launches (round 5)
hipError_t launch_fused2d_inv_2k2k(int twd, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw16r, const int2 *h_tw, const uint32_t *tw2d, size_t nframes,
                                   hipStream_t stream); // ... and its inverse
int fused2d_inv_supported(int log2n, int l1, int data_width, int twdl_width, int format, int rndmode, int direction, int in_order, int out_order); // 1: N = 2^20, 2: N = 2^21, 3: three launches, 4: N = 2^22 = 2048 x 2048
hipError_t launch_fused2d_inv_cols(int l2, int twd, const uint32_t *rows, uint32_t *pout, const uint2 *tw1k, const int2 *h_tw1k, const uint32_t *tw2d, size_t nframes, int halves,
                                   hipStream_t stream);
hipError_t launch_fused2d_inv21(int twd, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw1k, const int2 *h_tw1k, const uint2 *tw16r, const int2 *h_tw2k,
                                const uint32_t *tw2d, size_t nframes, int halves, hipStream_t stream);
hipError_t launch_fused2d_inv(int twd, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw1k, const int2 *h_tw1k, const uint32_t *tw2d, size_t nframes,
                              int halves, hipStream_t stream);
const char *fused2d_kernel_name();
hipError_t launch_fused2d(int twd, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw1k, const int2 *h_tw1k, const uint32_t *tw2d, size_t nframes,
                          int halves, hipStream_t stream);
hipError_t launch_fused2d_rows2k(int twd, const uint32_t *prod, uint32_t *pout, const uint2 *tw16r, const int2 *h_tw, size_t nframes, hipStream_t stream);
hipError_t launch_fused2d_cols(int lr, int twd, const uint32_t *pin, uint32_t *scr, const uint2 *tw1k, const int2 *h_tw1k, const uint32_t *tw2d, size_t nframes,
                               int halves, hipStream_t stream);
hipError_t launch_big2x_inv(int log2n, bool fx, const uint32_t *pin, uint32_t *pout, uint32_t *scr, const uint2 *tw16f, const int2 *h_tw, size_t nframes,
                            const struct Slice &sl, int halves, hipStream_t stream, bool in_bitrev = false);
// two-pass plans for N = 2^17, 2^18 forward: 32-register first pass (intfft_big2p.hip) + k_mid_p2 / k_mid_c
bool big2p_supported(int log2n);
bool big2p_tables_ok(int log2n, const int2 *h_tw, int twd);
hipError_t launch_big2p_a(int log2n, bool fx, const uint32_t *pin, uint32_t *scr, const uint2 *tw16f, size_t nframes, const struct Slice &sl,
                          int halves, hipStream_t stream);
hipError_t launch_big2p_q(int log2n, bool fx, const uint32_t *scr, uint32_t *pout, const uint2 *tw16f, size_t nframes, const struct Slice &sl,
                          int halves, hipStream_t stream);
hipError_t launch_fast4096_mid(int twd, void *scratch, size_t nblocks4k, const int2 *tw_all, const int2 *h_tw, hipStream_t stream,
                               int data_width = 16, int rndmode = 0);

// bit-permutation mover (intfft_reorder.hip): m_in bit in_of_out[b] = m_out bit b; frames of 2^L (re, im) container pairs
hipError_t launch_bitperm(int L, int container_bytes, const int *in_of_out, const void *d_in, void *d_out, size_t batch,
                          hipStream_t stream);
hipError_t launch_bitperm_tw(int L, int container_bytes, const int *in_of_out, int l2, int mw, int sh_a, int sh_b, int narrow, int twd,
                             int conj, const void *d_in, void *d_out, size_t batch, hipStream_t stream);
int order_mem_bit(int order, int L, int j); // memory-index bit that carries logical-index bit j in an INTFFT_ORDER_* layout
// USE_FLY = 0: n scalars wrapped to dw bits (sign- or zero-extended) from in_cb- to out_cb-byte containers (intfft_reorder.hip)
hipError_t launch_convert(int in_cb, int out_cb, int dw, int zext, const void *in, void *out, size_t n, hipStream_t stream);
// 2-D scheme (intfft_generic.hip): in place V <- cmult(V, W_N^(k1 * n2)) on the [k1][n2] layout (conj: the inverse's swapped feed)
hipError_t launch_twmul(void *data, int container_bytes, int L, int l2, int mw, int sh_a, int sh_b, int narrow, int conj,
                        int twd, size_t nframes, hipStream_t stream);

// packed int16 block kernels for N = 8192 / 16384 in one pass, FWD / INV (intfft_fast16k.hip)
bool fast16k_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly, int in_order, int out_order);
bool fast16k_tables_ok(int log2n, const int2 *h_tw, int twd);
hipError_t launch_fast16k(int log2n, int direction, int twd, const void *in, void *out, const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream,
                          int data_width, int rndmode = 0, int native_orders = 0); // native_orders: bit 0 HALVES on the time side, bit 1 BITREV on the frequency side
const char *fast16k_kernel_name();
// packed int16 block kernel for N = 4096, FWD / INV / PAIR (intfft_fast4096.hip)
bool fast4096_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly,
                        int in_order, int out_order);
hipError_t launch_fast4096(int log2n, int direction, int twd, int lc_bitrev, int halves, const void *in, void *out, const int2 *tw_all, const int2 *h_tw,
                           size_t nframes, hipStream_t stream, int round = 0, int data_width = 16);
const char *fast4096_kernel_name();

} // namespace intfft
