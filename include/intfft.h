/*
 * intfft.h -- C-ABI of the MI355X-native integer FFT/IFFT engine (libintfft.so).
 *
 * Drop-in boundary for the hot path of hukenovs/intfftk: the reference has no software API, so
 * the boundary is the RTL entity interface of int_fftNk / int_ifftNk
 * (src/vhdl/fft/int_fftNk.vhd:72-103, src/vhdl/fft/int_ifftNk.vhd:71-102) and of the two
 * wrappers that call them (src/vhdl/main/int_fft_single_path.vhd:85-113,
 * src/vhdl/main/int_fft_ifft_pair.vhd:74-107).  Generics become `intfft_params`; the 2-lane
 * valid-qualified sample stream becomes a frame-major device array; one RTL frame (N/2 beats)
 * becomes one row of `batch`.
 *
 * Plain C: no torch types, no C++ types, no exceptions across the boundary.
 * Data pointers are HIP device pointers; the caller owns them.
 */
#ifndef INTFFT_H
#define INTFFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: 0 OK, negative = parameter/elaboration errors, positive = hipError_t ---- */
#define INTFFT_OK               0
#define INTFFT_ERR_INVALID     (-1) /* parameter out of range (e.g. log2n, orders, direction)         */
#define INTFFT_ERR_UNSUPPORTED (-2) /* widths for which the RTL does not elaborate: find_delay -> 0,
                                       int_dif2_fly.vhd:87-116; a trpl18 multiplier whose product slice
                                       leaves P (MAW + MBW > 80 | 78, int_cmult_trpl18_dsp48.vhd:151-152);
                                       2-D scheme plans with results wider than 64 bits              */
#define INTFFT_ERR_NULL        (-3) /* NULL argument                                                  */
#define INTFFT_ERR_NO_DEVICE   (-4) /* no HIP device / wrong device index -- there is NO CPU fallback */
#define INTFFT_ERR_ALLOC       (-5) /* host allocation failed                                         */
#define INTFFT_ERR_TRANSPORT   (-6) /* intfft_shard_set_transport / intfft_exec_sharded: RCCL cannot be loaded, two
                                       plans of the set share a device, or an RCCL call failed          */

/* direction: which core(s) a frame goes through */
#define INTFFT_FWD  0 /* int_fftNk  : radix-2 DIF forward  (int_fftNk.vhd:72)                     */
#define INTFFT_INV  1 /* int_ifftNk : radix-2 DIT inverse  (int_ifftNk.vhd:71)                    */
#define INTFFT_PAIR 2 /* int_fftNk -> int_ifftNk back to back (int_fft_ifft_pair.vhd:209-280)     */

/*
 * I/O orders: how memory index m of a frame maps to the logical index (time index n for the
 * input of FWD / output of INV; frequency index k for the output of FWD / input of INV).
 *   NATURAL      : logical = m                    (int_fft_single_path.vhd:42-47 serial I/O;
 *                                                  == the interleave-2 stream of iobuf_flow_int2.vhd:18-40)
 *   BITREV       : logical = bitrev(m)            native int_fftNk output / int_ifftNk input as
 *                                                  beats (lane0[i], lane1[i]) = (m = 2i, 2i+1)
 *                                                  (int_fftNk.vhd:18-21, math/fn_radix2.m:182-188)
 *   HALVES       : logical = (m >> 1) + (m & 1) * N/2   native int_fftNk input / int_ifftNk output
 *                                                  as beats (lane0[i], lane1[i]) (int_fftNk.vhd:15-17)
 *   BITREV_LANES : logical = bitrev(2 * (m mod N/2) + m div N/2)   the serial stream
 *                                                  [lane0 frame ; lane1 frame] of outbuf_half_path.vhd:160-172,
 *                                                  i.e. what int_bitrev_order.vhd:82-104 turns into NATURAL
 */
#define INTFFT_ORDER_NATURAL      0
#define INTFFT_ORDER_BITREV       1
#define INTFFT_ORDER_HALVES       2
#define INTFFT_ORDER_BITREV_LANES 3

/* Mirrors the VHDL generics 1:1 (int_fftNk.vhd:73-84; tb mode names fft_signle_test.vhd:80-112:
 * "UNSCALED" = format 1, "TRUNCATE" = format 0 rndmode 0, "ROUNDING" = format 0 rndmode 1). */
typedef struct intfft_params {
    int32_t log2n;      /* NFFT      : log2 of the length, 3..19 (20 = documented extension)      */
    int32_t data_width; /* DATA_WIDTH: width of the input samples, 2..64 (wrappers use 8..32)      */
    int32_t twdl_width; /* TWDL_WIDTH: 4..27 (XSER NEW) / 4..25 (OLD)                              */
    int32_t format;     /* FORMAT    : 1 unscaled (1 bit growth per stage), 0 scaled              */
    int32_t rndmode;    /* RNDMODE   : 0 truncate, 1 round-half-up; scaled only                   */
    int32_t xser;       /* XSER      : 0 "OLD" (DSP48E1), 1 "NEW" (DSP48E2) -- changes results in
                                       the wide-multiplier and Taylor-twiddle regimes              */
    int32_t direction;  /* INTFFT_FWD / INTFFT_INV / INTFFT_PAIR                                   */
    int32_t use_fly;    /* USE_FLY   : 1 normal, 0 bypass butterflies (int_fftNk.vhd:260-277)      */
    int32_t in_order;   /* INTFFT_ORDER_*                                                          */
    int32_t out_order;  /* INTFFT_ORDER_*                                                          */
} intfft_params;

typedef struct intfft_plan intfft_plan;

/* What a plan resolved to; for benchmarks and tests. */
typedef struct intfft_plan_info {
    int32_t in_bits, out_bits;           /* DATA_WIDTH, DATA_WIDTH + FORMAT*NFFT (x2 for PAIR)     */
    int32_t in_container, out_container; /* bytes per real component: 2, 4, 8 (16 out: see below)  */
    int32_t n_passes;                    /* kernel launches per batch chunk                        */
    int32_t compute_word;                /* bytes of the on-chip word (2 = packed int16 kernels)   */
    int32_t fast_path;                   /* 1 if a single dedicated kernel serves this plan        */
    int32_t reserved;
    uint64_t scratch_bytes;              /* plan-owned device scratch (0 after intfft_plan_release_scratch) */
    char kernel_name[64];                /* dominant kernel symbol (for rocprof matching)          */
} intfft_plan_info;

/* Widths and containers implied by the generics.  Containers are the smallest of int16/32/64
 * that hold the width; frames are [batch][N] of interleaved (re, im), sign-extended.  Input
 * values outside data_width are wrapped on load like conv_std_logic_vector in
 * fft_signle_test.vhd:163-164.  Results wider than 64 bits (bit growth into the trpl18 multiplier
 * tail, int_cmult_dsp48.vhd:267-303: at most 74 bits) come in 16-byte containers: one little-endian
 * two's-complement 128-bit integer per component (low 64-bit word first), 32 bytes per sample.
 * Inputs never need them (data_width <= 64). */
int intfft_io_widths(const intfft_params *p, int *in_bits, int *out_bits, int *in_container_bytes,
                     int *out_container_bytes);

/* Elaborates a core on a HIP device: validates the generics the way the RTL elaboration would,
 * generates the twiddle tables on the device (rom_twiddle_int.vhd + row_twiddle_tay.vhd) and
 * chooses the kernels.  intfft_exec never modifies the plan; intfft_exec_host, intfft_shard_prepare and
 * intfft_exec_sharded create (grow-only) staging state inside it on first use -- see their comments. */
int intfft_plan_create(intfft_plan **out, const intfft_params *p, int hip_device);
/* N > 512K: the "2D-FFT scheme" the reference names but does not define (int_fftNk.vhd:11-13: "For N > 512K you
 * should use 2D-FFT scheme"; row_twiddle_tay.vhd:31).  THIS IS AN EXTENSION OF THIS LIBRARY, specified in DESIGN.md
 * section 4.5 and restated in oracle/: N = 2^log2n = N1 * N2 with N1 = 2^log2_n1 and both factors native core lengths
 * (3..19 bits, log2n <= 24), built from the reference's own blocks --
 *   forward  N1-point int_fftNk over n1 (n = n1*N2 + n2) for every n2; one int_cmult_dsp48 per sample by the
 *            inter-pass twiddle W_N^(k1*n2) (the quarter-wave ROM formula of rom_twiddle_int.vhd:143-152 at full
 *            depth, no Taylor step) at the width reached there; N2-point int_fftNk over n2 for every k1; X[k1 +
 *            N1*k2]
 *   inverse  the mirror with int_ifftNk and the re/im-swapped multiplier feed (int_dit2_fly.vhd:304-322);
 *            pair = both.
 * Widths, scaling, rounding, XSER, orders and containers mean what they mean for the 1-D cores (DATA_WIDTH +
 * FORMAT*log2n bits out); `intfft_params.log2n` is the TOTAL length, so the ABI struct is unchanged.  Results differ
 * in the last bits from a 1-D plan of the same length (different twiddle factorisation); both are within the same
 * distance of the exact DFT.  use_fly = 0 is not defined for this scheme (INTFFT_ERR_INVALID).  intfft_twiddles(plan,
 * -1, ..) returns the inter-pass table (N entries), stages 0 .. max(log2 N1, log2 N2) - 1 the per-stage tables shared
 * by the two cores. */
int intfft_plan_create_2d(intfft_plan **out, const intfft_params *p, int log2_n1, int hip_device);
int intfft_plan_destroy(intfft_plan *plan);
int intfft_plan_get_info(const intfft_plan *plan, intfft_plan_info *info);

/* Transforms `batch` frames: d_in/d_out are device pointers to [batch][N][2] containers (see intfft_io_widths),
 * aligned to one complex sample (2 containers: 4 bytes for int16 pairs) -- the dedicated kernels then issue 8- and
 * 16-byte vector accesses on addresses that are only sample-aligned, which is legal in the unaligned-access mode HIP
 * runs gfx950 in (SH_MEM_CONFIG.alignment_mode = unaligned, the ROCm default; tests/test_gpu_cabi.py shifts every
 * kernel family's buffers by one sample); buffers from hipMalloc are 256-byte aligned and need no thought; nothing
 * outside the output array is written and the loads of absent frames of a partial last group are predicated
 * (tests/test_gpu_cabi.py::test_no_writes_outside_the_output_buffer: guard bands, ragged batches, buffers one sample
 * off a 64 KiB boundary). Asynchronous on `hip_stream` (a hipStream_t, NULL = default stream). d_in == d_out is
 * allowed when the containers have equal size; any other overlap of the two byte ranges returns INTFFT_ERR_INVALID (a
 * block would overwrite frames another block has not read).  Re-entrant across plans (launch geometry is cached per
 * (kernel, device) on first use, under a mutex); a plan WITHOUT plan-owned scratch (intfft_plan_info.scratch_bytes ==
 * 0: every single-launch plan) holds no mutable state and may be executed on any number of streams at once; a plan
 * that owns scratch (the multi-pass plans) must not be executed concurrently on two streams THROUGH THIS ENTRY POINT
 * -- use intfft_exec_ws below with one workspace per stream (or one plan per stream).  Some multi-pass plans (N =
 * 2^19 / 2^20 forward and inverse, the 24-bit unscaled class, the tiled 2-D plans) run the scratch-sized chunks of a
 * large batch alternately on `hip_stream` and on a side stream taken from a pool inside the plan for the duration of
 * the call (event fork at entry, event join before returning): towards the caller the call is still ordered on
 * `hip_stream` only. */
int intfft_exec(intfft_plan *plan, const void *d_in, void *d_out, size_t batch, void *hip_stream);

/* The same transform on a CALLER-SUPPLIED workspace: every plan is re-entrant through this entry point.  The RTL core
 * is stateless between instances (int_fftNk.vhd:23-37: frames may follow each other back to back, nothing is kept
 * between them) and SURVEY.md section 8(b) asks for an exec that is re-entrant across streams: with the inter-pass
 * scratch, the layout buffers of the 2-D plans and the middle buffer of composite pairs all carved out of
 * `d_workspace`, a call reads the plan and writes nothing in it (the side stream of the two-stream plans comes from a
 * mutex-protected pool and goes back when the call returns), so ONE plan may run on any number of streams / host
 * threads at once as long as every concurrent call has its own workspace.
 *   intfft_plan_workspace_bytes(plan, batch, &bytes): the workspace with which a call of `batch` frames runs exactly
 *     like intfft_exec on the plan's own scratch (0 for single-launch plans: d_workspace may then be NULL).  Monotone
 *     in `batch` and bounded, but the bound depends on the plan family -- two 128 MiB scratch halves for the 1-D
 *     multi-pass plans; one or two 256 MiB layout buffers PLUS the row sub-plan's share for the 2-D scheme plans (512
 *     MiB and more); the middle buffer plus the larger sub-plan for composite pairs -- so size pools from THIS call
 *     (e.g. with batch = SIZE_MAX for the largest a plan ever asks for), never from a constant.
 *   intfft_exec_ws(..., d_workspace, ws_bytes, stream): d_workspace is a device pointer on the plan's device,
 *     256-byte aligned, not overlapping d_in / d_out.  A workspace smaller than intfft_plan_workspace_bytes(plan,
 *     batch) is accepted as long as it serves one frame (>= intfft_plan_workspace_bytes(plan, 1)): the batch is then
 *     cut into the largest sub-batches the workspace serves, one after the other on `hip_stream` (slower: no
 *     two-stream overlap inside a sub-batch that fits one scratch half); smaller than that -> INTFFT_ERR_INVALID.
 *   intfft_plan_release_scratch(plan): frees the plan-owned scratch (and that of its sub-plans) after waiting for the
 *     device; from then on intfft_exec / intfft_exec_host / intfft_exec_sharded on this plan return
 *     INTFFT_ERR_INVALID and only intfft_exec_ws runs it -- for callers that bring their own workspaces and do not
 *     want N x up to 256 MiB held by plans. */
int intfft_plan_workspace_bytes(const intfft_plan *plan, size_t batch, size_t *bytes);
int intfft_exec_ws(intfft_plan *plan, const void *d_in, void *d_out, size_t batch, void *d_workspace, size_t ws_bytes,
                   void *hip_stream);
int intfft_plan_release_scratch(intfft_plan *plan);

/* Host-resident frames (the "streaming block" use): h_in/h_out are HOST pointers with the same layout
 * as above.  The batch is cut into chunks of `chunk_frames` frames (0 = library default) that are
 * staged through two device slots and three HIP streams, so that the upload of chunk i+1, the
 * transform of chunk i and the download of chunk i-1 overlap -- the software analogue of feeding
 * the core frames back to back (int_fftNk.vhd:23-37).  Blocking; returns when h_out is complete.
 * Pinned buffers (hipHostMalloc, or registered by their owner) make the copies truly asynchronous; pageable buffers
 * work too (the runtime stages them: less overlap).  The library never registers the caller's memory itself.
 * This is NOT a CPU execution path: every frame is transformed on the HIP device. */
int intfft_exec_host(intfft_plan *plan, const void *h_in, void *h_out, size_t batch, size_t chunk_frames);

/* The frame-queue form of the same interface: frames arrive OVER TIME (1 .. any number per call, with gaps), results
 * leave in the order the frames went in -- what the RTL core does with its valid strobes (int_fftNk.vhd:23-37), and
 * what SURVEY.md section 8(f) N2 asks for ("frames arriving in chunks, double-buffered").  Unlike intfft_exec_host
 * the overlap of upload, transform and download works ACROSS calls.
 *   intfft_stream_open(plan, slot_frames, n_slots, &s)
 *       A stream object on the plan's device: n_slots (0 = 3; 2 .. 64) slots of slot_frames frames (0 = about 32
 *       MiB), each a pinned host input buffer, device input / output buffers and a pinned host output buffer owned by
 *       the library, three HIP streams, and a private workspace (intfft_plan_workspace_bytes(plan, slot_frames)).
 *       Transforms run through intfft_exec_ws: the plan is only read, so ONE plan may feed any number of stream
 *       objects at once, also after intfft_plan_release_scratch.  The plan must outlive the stream object.
 *   intfft_stream_push(s, h_frames, nframes, &accepted)
 *       Copies frames (host pointer, any memory: pageable is fine -- the library's pinned ring is what the DMA engine
 *       reads) into the filling slot; every slot that becomes full is submitted (upload -> transform -> download,
 *       each on its own HIP stream, ordered by events).  NEVER waits for the device: when all slots are in flight or
 *       hold results that were not pulled yet it returns INTFFT_OK with *accepted < nframes (accepted may be NULL).
 *   intfft_stream_flush(s)
 *       Submits the partly filled slot as a short chunk (a producer that pauses, or the end of the data).  Producer
 *       side.
 *   intfft_stream_pull(s, h_out, max_frames, &got, wait)
 *       Copies up to max_frames finished frames, in push order, into h_out.  wait = 0: only what is complete now (got
 *       may be 0); wait = 1: if something was submitted and is not pulled yet, blocks until at least its oldest slot
 *       is complete.  Frames that sit in a slot that was not submitted (not full, not flushed) are not waited for:
 *       got = 0.
 *   intfft_stream_pending(s, &not_pulled, &not_submitted)
 *       Frames pushed and not pulled yet / frames in the filling slot.
 *   intfft_stream_close(s)   drains the device work, drops results that were not pulled, frees everything.
 * Threading: ONE producer thread (push, flush) and ONE consumer thread (pull) may work on a stream object
 * concurrently. Errors are sticky: after a failed enqueue every call on the object returns that status (hipError_t >
 * 0 or INTFFT_ERR_* < 0) until close.  This is NOT a CPU execution path either. */
typedef struct intfft_stream intfft_stream;
int intfft_stream_open(intfft_plan *plan, size_t slot_frames, int n_slots, intfft_stream **out);
int intfft_stream_push(intfft_stream *s, const void *h_frames, size_t nframes, size_t *accepted);
int intfft_stream_flush(intfft_stream *s);
int intfft_stream_pull(intfft_stream *s, void *h_out, size_t max_frames, size_t *got, int wait);
int intfft_stream_pending(intfft_stream *s, size_t *frames_not_pulled, size_t *frames_not_submitted);
int intfft_stream_close(intfft_stream *s);

/* Single-process multi-GPU convenience (frames are independent, so a batch shards across
 * GPUs with no collective in the data path -- the software analogue of instantiating the core once per channel).
 * plans[0..nplans-1] hold identical intfft_params, one per HIP device; d_in / d_out live on the device of
 * plans[root].  The batch is cut into contiguous shards (remainder to the LAST plans); shard i is copied root ->
 * device i (hipMemcpyPeerAsync over xGMI, peer access enabled where the devices allow it), transformed there, and
 * copied back; the root transforms its own shard in place of the copy.  Blocking. Every peer's shard moves in up to 4
 * pieces (>= 2 MiB each) so that the copy back of piece k runs beside the copy in of piece k + 1 (xGMI links are full
 * duplex; end to end this path is link-bound) and beside the transform of the root's own shard.
 * Synchronisation contract: on entry the call waits for EVERY stream of the root device (hipDeviceSynchronize), so
 * d_in may have been produced on any stream of that device; on return d_out is complete.  On an error every stream
 * the call used is still drained before it returns. Plan state: the per-shard staging buffers and stream live in the
 * plans (grow only, freed by intfft_plan_destroy) and are created on first use -- or up front by
 * intfft_shard_prepare, after which intfft_exec_sharded allocates nothing for batches <= max_batch.  These two calls
 * are the only ones besides intfft_exec_host that modify a plan after create: do not run them concurrently with any
 * other call on the same plans. One-process-per-GPU hosts (torch.distributed / MPI over RCCL) call intfft_exec on
 * their own shard instead (intfftk_amd/sharding.py: grouped ncclSend/ncclRecv scatter and gather). */
int intfft_shard_prepare(intfft_plan *const *plans, int nplans, int root, size_t max_batch);
int intfft_exec_sharded(intfft_plan *const *plans, int nplans, int root, const void *d_in, void *d_out, size_t batch);
/* The same call without host synchronisation -- a "streaming block" over several GPUs that can be enqueued behind
 * other work: everything is ordered behind what `hip_stream` (a stream of plans[root]'s device, NULL = its default
 * stream) holds at the time of the call (d_in must be complete in THAT stream's order -- no hipDeviceSynchronize
 * here), and when the call returns `hip_stream` has been made to wait for every copy and transform of the call, so
 * work enqueued on it afterwards sees d_out complete.  Back-to-back calls on the same plans are ordered among
 * themselves (each plan's streams first wait for the previous call's completion event).  Staging that has to grow is
 * (re)allocated inside the call, which synchronises the device: call intfft_shard_prepare first.  Same state and
 * threading rules as intfft_exec_sharded. */
int intfft_exec_sharded_async(intfft_plan *const *plans, int nplans, int root, const void *d_in, void *d_out, size_t batch,
                              void *hip_stream);
/* How intfft_exec_sharded moves the shards between the root and the other devices ("grouped
 * ncclSend / ncclRecv root <-> peers so that all links of the root are driven concurrently"):
 *   INTFFT_TRANSPORT_PEER  hipMemcpyPeerAsync per shard on the shard's stream (the default)
 *   INTFFT_TRANSPORT_RCCL  EXPERIMENTAL -- no run on two or more devices has been recorded for it yet (every box this
 *                          library was measured on had ONE GPU: with one rank the groups are empty, so only dlopen,
 *                          ncclCommInitAll and the empty-group path are proven; tests/test_gpu_cabi.py compares it
 *                          with the peer transport whenever >= 2 devices are visible). RCCL over xGMI: per piece of
 *                          the shards ONE ncclGroupStart .. ncclGroupEnd holding the ncclSend (root) / ncclRecv
 *                          (peer) pairs of the scatter of piece t AND the reverse pairs of the gather of piece t - 1,
 *                          so that every link of the root runs in both directions at once.  librccl.so is loaded with
 *                          dlopen on the first request (no link-time dependency); the communicators (ncclCommInitAll
 *                          over the plans' devices, rank i = plans[i]) belong to the plan set and are released by
 *                          intfft_plan_destroy of plans[0].  Needs every plan on its own device.
 * Returns INTFFT_OK; INTFFT_ERR_TRANSPORT when RCCL cannot be loaded or the communicators cannot be created (two
 * plans on one device, no librccl.so): the plan set then stays on peer copies, so a caller may simply try RCCL first.
 * The two transports give the same bytes.  A set is identified by the call that created it: plans whose communicators
 * come from different calls (or whose owner, plans[0] of that call, was destroyed or re-assigned since) fall back to
 * peer copies rather than touching a stale communicator.  Like intfft_shard_prepare this call modifies the plans: not
 * concurrently with any other call on them. */
#define INTFFT_TRANSPORT_PEER 0
#define INTFFT_TRANSPORT_RCCL 1
int intfft_shard_set_transport(intfft_plan *const *plans, int nplans, int root, int transport);

/* Standalone re-orderer: the stream buffers of src/vhdl/buffers/ as an operator of their own (no plan, no
 * arithmetic). d_out[f][m_out] = d_in[f][m_in] for every frame f, where m_in (memory index in `from_order`) and m_out
 * (memory index in `to_order`) denote the same logical index -- e.g. int_bitrev_order.vhd:61-189 = BITREV_LANES ->
 * NATURAL, inbuf_half_path.vhd:23-28 = NATURAL -> HALVES, outbuf_half_path.vhd:160-172 = BITREV -> BITREV_LANES, and
 * NATURAL -> BITREV = the bitrevorder() of math/fn_radix2.m:188.  Samples are (re, im) pairs of `container_bytes` (2
 * / 4 / 8) each; frames are [batch][2^log2n].  Not in place (overlapping buffers -> INTFFT_ERR_INVALID). Asynchronous
 * on `hip_stream` of device `hip_device`. */
int intfft_reorder(int log2n, int container_bytes, int from_order, int to_order, const void *d_in, void *d_out,
                   size_t batch, int hip_device, void *hip_stream);

/* Parity introspection: the twiddle stream of butterfly stage `stage` (2^stage entries, the
 * values rom_twiddle_int emits for cnt = 0 .. 2^stage-1) as interleaved int32 (re, im).
 * h_out may be NULL to query *count. */
int intfft_twiddles(const intfft_plan *plan, int stage, int32_t *h_out, size_t *count);

/* Environment: NONE in normal use.  The library has DIAGNOSTIC switches (A/B parity of kernel families in the tests,
 * tuning experiments; none changes results) that are read ONLY when the master switch INTFFT_DIAG=1 is set -- without
 * it every INTFFT_* variable is ignored, so a stray variable in a production environment cannot change which kernels
 * a plan uses:
 *   INTFFT_GENERIC_ONLY, INTFFT_NO_FAST1024U, INTFFT_NO_FASTW32, INTFFT_NO_BIG20, INTFFT_NO_BIG2P, INTFFT_NO_BIG2X,
 *   INTFFT_NO_FAST16K, INTFFT_TWO_STREAMS, INTFFT_NO_WIDE16, INTFFT_NO_WIDELONG, INTFFT_NO_WIDELONG_R32,
 *   INTFFT_NO_BIGWLONG (the three-launch plans of N = 2^17 .. 2^20 outside 16-bit scaled data: back to the generic
 *   passes), INTFFT_NO_FASTW64, INTFFT_NO_PAIR_COMPOSITE, INTFFT_NO_LANES_COMPOSITE, INTFFT_NO_BYPASS_COPY,
 *   INTFFT_NO_ROTATE1, INTFFT_NO_NARROW_PASS, INTFFT_NO_TWOPASS, INTFFT_NO_PACKED_ROUND, INTFFT_NO_NARROW16,
 *   INTFFT_2D_GENERIC, INTFFT_2D_NO_FUSE, INTFFT_2D_NO_FUSED_CORES, INTFFT_2D_NO_ROWS2K, INTFFT_2D_NO_PACKED_TW,
 *   INTFFT_2D_CHUNK_FRAMES (which kernels a plan may use), INTFFT_FAST_EXTRACT, INTFFT_FAST_PIPE (code paths inside
 *   the packed kernels), INTFFT_BLOCKS_PER_CU, INTFFT_SCRATCH_MB, INTFFT_ONE_STREAM, INTFFT_TILE_LOG2,
 *   INTFFT_PASS_THREADS, INTFFT_PASS_TARGET, INTFFT_NO_MIXED_WORDS, INTFFT_NO_NARROW_MUL, INTFFT_2XA_HALF19,
 *   INTFFT_2XA_FULL20, INTFFT_SHARD_PIECES (launch geometry / scratch / generic-kernel / shard-pipeline knobs),
 *   INTFFT_VERBOSE (the failing RCCL call and its ncclResult_t on stderr).  README.md describes each. */
const char *intfft_strerror(int status);
const char *intfft_version(void);

#ifdef __cplusplus
}
#endif
#endif /* INTFFT_H */
