// intfft_stream.hip -- the frame-queue form of the streaming host interface (include/intfft.h: intfft_stream_*).
//
// The RTL core takes frames as they arrive -- back to back or with gaps between them (src/vhdl/fft/int_fftNk.vhd:23-37: RAMB_TYPE "WRAP"
// tolerates a wrapped strobe, "CONT" needs continuous frames) -- and hands results out in the order the frames went in.  This is that
// contract for a host producer: push frames whenever they exist, pull results whenever they are ready; the upload of one slot, the
// transform of the previous one and the download of the one before overlap ACROSS calls, which the blocking intfft_exec_host (a batch
// that is already complete in host memory) cannot give.
//
//   slot  = { pinned host input, device input, device output, pinned host output, three events }      n_slots of them, used round robin
//   push  copies frames into the filling slot's pinned input; a full slot is submitted:  H2D (stream up) -> intfft_exec_ws (stream
//         comp, behind the upload's event) -> D2H (stream down, behind the transform's event).  Never blocks on the device: when every
//         slot is in flight or waiting to be pulled it returns with *accepted < nframes.
//   flush submits the partly filled slot (a short chunk), so that a producer that pauses gets its last frames out.
//   pull  hands out finished frames in push order from the oldest slot's pinned output; a slot is free again once all its frames are pulled.
//
// Every transform runs through intfft_exec_ws on a workspace owned by the stream object: the plan is only read, so one plan may feed any
// number of stream objects (and other callers) at once.  One producer thread and one consumer thread may use a stream object
// concurrently (a mutex guards the slot states; the bulk memcpys run outside it -- a slot is touched by the producer only while it is
// filling and by the consumer only while it is draining).  This is NOT a CPU execution path: every frame is transformed on the HIP device.
#include "intfft_internal.hpp"

#include "../../include/intfft.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace intfft {
void plan_geometry(const intfft_plan *plan, int *device, int *log2n, int *in_cb, int *out_cb); // intfft_plan.hip
}

namespace {

enum SlotState : int { SLOT_FREE = 0, SLOT_FILLING = 1, SLOT_INFLIGHT = 2 };

struct Slot {
    void *h_in = nullptr, *h_out = nullptr; // pinned
    void *d_in = nullptr, *d_out = nullptr;
    hipEvent_t ev_up = nullptr, ev_comp = nullptr, ev_down = nullptr;
    size_t fill = 0;  // frames written by push (FILLING)
    size_t count = 0; // frames submitted (INFLIGHT)
    size_t read = 0;  // frames already pulled (INFLIGHT)
    int state = SLOT_FREE;
};

struct DevGuard {
    int prev = -1;
    bool ok = false;
    explicit DevGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

} // namespace

struct intfft_stream {
    intfft_plan *plan = nullptr;
    int device = 0;
    size_t in_frame = 0, out_frame = 0; // bytes per frame
    size_t slot_frames = 0;
    std::vector<Slot> slots;
    hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
    void *d_ws = nullptr;
    size_t ws_bytes = 0;
    std::mutex mu;
    unsigned long long push_seq = 0, pull_seq = 0; // slot = seq % n_slots
    unsigned long long frames_in = 0, frames_out = 0;
    int error = INTFFT_OK; // sticky: the first failure of an enqueue / transform
};

static void stream_free(intfft_stream *s)
{
    for (Slot &sl : s->slots) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
        if (sl.ev_comp) (void)hipEventDestroy(sl.ev_comp);
        if (sl.ev_down) (void)hipEventDestroy(sl.ev_down);
    }
    if (s->d_ws) (void)hipFree(s->d_ws);
    if (s->s_up) (void)hipStreamDestroy(s->s_up);
    if (s->s_comp) (void)hipStreamDestroy(s->s_comp);
    if (s->s_down) (void)hipStreamDestroy(s->s_down);
    delete s;
}

// enqueue upload -> transform -> download of a slot holding `fill` frames (caller holds the mutex and has set the device)
static int submit(intfft_stream *s, Slot &sl)
{
    const size_t nf = sl.fill;
    hipError_t e = hipMemcpyAsync(sl.d_in, sl.h_in, nf * s->in_frame, hipMemcpyHostToDevice, s->s_up);
    if (e == hipSuccess) e = hipEventRecord(sl.ev_up, s->s_up);
    if (e == hipSuccess) e = hipStreamWaitEvent(s->s_comp, sl.ev_up, 0);
    int rc = INTFFT_OK;
    if (e == hipSuccess) rc = intfft_exec_ws(s->plan, sl.d_in, sl.d_out, nf, s->d_ws, s->ws_bytes, s->s_comp);
    if (e == hipSuccess && rc == INTFFT_OK) e = hipEventRecord(sl.ev_comp, s->s_comp);
    if (e == hipSuccess && rc == INTFFT_OK) e = hipStreamWaitEvent(s->s_down, sl.ev_comp, 0);
    if (e == hipSuccess && rc == INTFFT_OK) e = hipMemcpyAsync(sl.h_out, sl.d_out, nf * s->out_frame, hipMemcpyDeviceToHost, s->s_down);
    if (e == hipSuccess && rc == INTFFT_OK) e = hipEventRecord(sl.ev_down, s->s_down);
    if (rc == INTFFT_OK && e != hipSuccess) rc = (int)e;
    if (rc != INTFFT_OK) return rc;
    sl.count = nf, sl.read = 0, sl.fill = 0, sl.state = SLOT_INFLIGHT;
    ++s->push_seq;
    return INTFFT_OK;
}

extern "C" {

int intfft_stream_open(intfft_plan *plan, size_t slot_frames, int n_slots, intfft_stream **out)
{
    if (!plan || !out) return INTFFT_ERR_NULL;
    *out = nullptr;
    if (n_slots == 0) n_slots = 3;
    if (n_slots < 2 || n_slots > 64) return INTFFT_ERR_INVALID;
    int device = 0, log2n = 0, in_cb = 0, out_cb = 0;
    intfft::plan_geometry(plan, &device, &log2n, &in_cb, &out_cb);
    DevGuard guard(device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    intfft_stream *s = new (std::nothrow) intfft_stream;
    if (!s) return INTFFT_ERR_ALLOC;
    s->plan = plan, s->device = device;
    s->in_frame = ((size_t)2 << log2n) * (size_t)in_cb, s->out_frame = ((size_t)2 << log2n) * (size_t)out_cb;
    if (slot_frames == 0) slot_frames = std::max<size_t>(1, ((size_t)32 << 20) / std::max(s->in_frame, s->out_frame));
    s->slot_frames = slot_frames;
    s->slots.resize((size_t)n_slots);
    hipError_t e = hipStreamCreateWithFlags(&s->s_up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->s_comp, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->s_down, hipStreamNonBlocking);
    for (Slot &sl : s->slots) {
        if (e == hipSuccess) e = hipHostMalloc(&sl.h_in, slot_frames * s->in_frame, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(&sl.h_out, slot_frames * s->out_frame, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&sl.d_in, slot_frames * s->in_frame);
        if (e == hipSuccess) e = hipMalloc(&sl.d_out, slot_frames * s->out_frame);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_comp, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_down, hipEventDisableTiming);
    }
    int rc = INTFFT_OK;
    if (e == hipSuccess) rc = intfft_plan_workspace_bytes(plan, slot_frames, &s->ws_bytes);
    if (e == hipSuccess && rc == INTFFT_OK && s->ws_bytes) e = hipMalloc(&s->d_ws, s->ws_bytes);
    if (e != hipSuccess || rc != INTFFT_OK) {
        stream_free(s);
        return rc != INTFFT_OK ? rc : (int)e;
    }
    *out = s;
    return INTFFT_OK;
}

int intfft_stream_push(intfft_stream *s, const void *h_frames, size_t nframes, size_t *accepted)
{
    if (accepted) *accepted = 0;
    if (!s || (nframes && !h_frames)) return INTFFT_ERR_NULL;
    DevGuard guard(s->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    size_t done = 0;
    const char *src = static_cast<const char *>(h_frames);
    while (done < nframes) {
        Slot *sl;
        size_t at, take;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (s->error != INTFFT_OK) return s->error;
            sl = &s->slots[s->push_seq % s->slots.size()];
            if (sl->state == SLOT_INFLIGHT) break; // ring full: the consumer has to pull first
            if (sl->state == SLOT_FREE) sl->state = SLOT_FILLING, sl->fill = 0;
            at = sl->fill;
            take = std::min(nframes - done, s->slot_frames - at);
        }
        // the producer is the only one to touch a FILLING slot: the bulk copy runs outside the mutex
        std::memcpy(static_cast<char *>(sl->h_in) + at * s->in_frame, src + done * s->in_frame, take * s->in_frame);
        done += take;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            sl->fill = at + take;
            s->frames_in += take;
            if (sl->fill == s->slot_frames) {
                const int rc = submit(s, *sl);
                if (rc != INTFFT_OK) {
                    s->error = rc;
                    if (accepted) *accepted = done;
                    return rc;
                }
            }
        }
    }
    if (accepted) *accepted = done;
    return INTFFT_OK;
}

int intfft_stream_flush(intfft_stream *s)
{
    if (!s) return INTFFT_ERR_NULL;
    DevGuard guard(s->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->error != INTFFT_OK) return s->error;
    Slot &sl = s->slots[s->push_seq % s->slots.size()];
    if (sl.state == SLOT_FILLING && sl.fill > 0) {
        const int rc = submit(s, sl);
        if (rc != INTFFT_OK) return s->error = rc;
    }
    return INTFFT_OK;
}

int intfft_stream_pull(intfft_stream *s, void *h_out, size_t max_frames, size_t *got, int wait)
{
    if (got) *got = 0;
    if (!s || !got || (max_frames && !h_out)) return INTFFT_ERR_NULL;
    DevGuard guard(s->device);
    if (!guard.ok) return INTFFT_ERR_NO_DEVICE;
    size_t done = 0;
    char *dst = static_cast<char *>(h_out);
    while (done < max_frames) {
        Slot *sl;
        size_t at, take;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (s->error != INTFFT_OK) return s->error;
            sl = &s->slots[s->pull_seq % s->slots.size()];
            if (s->pull_seq == s->push_seq || sl->state != SLOT_INFLIGHT) break; // nothing submitted that is not pulled yet
        }
        // ev_down of an INFLIGHT slot was recorded under the mutex before the state changed; only the consumer moves pull_seq
        hipError_t e = (wait && done == 0) ? hipEventSynchronize(sl->ev_down) : hipEventQuery(sl->ev_down);
        if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            break;
        }
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> lk(s->mu);
            return s->error = (int)e;
        }
        at = sl->read;
        take = std::min(max_frames - done, sl->count - at);
        std::memcpy(dst + done * s->out_frame, static_cast<const char *>(sl->h_out) + at * s->out_frame, take * s->out_frame);
        done += take;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            sl->read = at + take;
            s->frames_out += take;
            if (sl->read == sl->count) {
                sl->state = SLOT_FREE, sl->count = sl->read = 0;
                ++s->pull_seq;
            }
        }
    }
    *got = done;
    return INTFFT_OK;
}

int intfft_stream_pending(intfft_stream *s, size_t *frames_not_pulled, size_t *frames_not_submitted)
{
    if (!s) return INTFFT_ERR_NULL;
    std::lock_guard<std::mutex> lk(s->mu);
    const Slot &cur = s->slots[s->push_seq % s->slots.size()];
    const size_t filling = cur.state == SLOT_FILLING ? cur.fill : 0;
    if (frames_not_pulled) *frames_not_pulled = (size_t)(s->frames_in - s->frames_out);
    if (frames_not_submitted) *frames_not_submitted = filling;
    return s->error;
}

int intfft_stream_close(intfft_stream *s)
{
    if (!s) return INTFFT_ERR_NULL;
    DevGuard guard(s->device);
    hipError_t e = hipSuccess;
    if (guard.ok) { // drain what is in flight; frames not pulled are dropped
        const hipError_t e1 = hipStreamSynchronize(s->s_up), e2 = hipStreamSynchronize(s->s_comp), e3 = hipStreamSynchronize(s->s_down);
        e = e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3;
    }
    stream_free(s);
    return guard.ok ? (int)e : INTFFT_ERR_NO_DEVICE;
}

} // extern "C"
