// intfft_fastw64.hip -- the 64-bit wave kernels at N = 1024 (every rounding kind and multiplier form) and their launcher; the kernels
// themselves are the templates of intfft_w64.hpp, the short-frame instances live in intfft_fastw64s.hip.
#include "intfft_w64.hpp"

namespace intfft {

bool fastw64_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order)
{
    const int out_bits = data_width + format * log2n;
    // N = 128 .. 1024 also in the cores' own beat orders (NAT instantiations, intfft_fastw64n.hip / intfft_fastw64sn.hip): int_fftNk HALVES in / BITREV out, int_ifftNk BITREV in / HALVES out
    const bool natural = in_order == 0 && out_order == 0;
    const bool native = log2n >= 7 && (direction == 0 ? (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1)
                                                       : (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2));
    return log2n >= 6 && log2n <= 10 && out_bits > 32 && out_bits <= 64 && data_width >= 2 && data_width <= 64 && (direction == 0 || direction == 1) && use_fly == 1 &&
           (natural || native) && twdl_width >= 4 && !diag_env("INTFFT_NO_FASTW64");
}

// what the multiplier stages 2 .. log2n - 1 of a plan allow (fly64<., ., CM>): 1 narrow, 3 three-dword products, 0 neither
int fastw64_multiplier_form(int log2n, const StageDesc *st10, int rnd_kind)
{
    bool narrow = true, three = true;
    for (int s = 2; s < log2n; ++s) {
        narrow = narrow && st10[s].narrow == 1;
        three = three && (st10[s].narrow == 0 || st10[s].narrow == 1) && st10[s].mw <= 63 && st10[s].sh_a + st10[s].sh_b <= 31;
    }
    // (round mode with three-dword products spills 180-260 dwords at 256 VGPRs: the general form is the faster one there)
    return narrow ? 1 : (three && rnd_kind != RND_ROUND) ? 3 : 0;
}

// N = 64 .. 512 are instantiated for the narrow multiplier form only (32-bit unscaled data with 16 .. 24-bit twiddles, scaled data up
// to 64 - TWDL_WIDTH bits); the planner sends the other short-frame plans to the generic kernel
bool fastw64_plan_ok(int log2n, const StageDesc *st10, int rnd_kind) { return log2n == 10 || fastw64_multiplier_form(log2n, st10, rnd_kind) == 1; }

// N = 2048 / 4096: the block kernels (intfft_fastw64b.hip / intfft_fastw64bi.hip)
bool fastw64b_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order)
{
    const int out_bits = data_width + format * log2n;
    return (log2n == 11 || log2n == 12) && out_bits > 32 && out_bits <= 64 && data_width >= 2 && data_width <= 64 && (direction == 0 || direction == 1) &&
           use_fly == 1 && twdl_width >= 4 && !diag_env("INTFFT_NO_FASTW64") &&
           (direction == 0 ? (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1) // NATURAL or the cores' own beat orders (intfft_fastw64bn.hip)
                           : (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2));
}
bool fastw64b_plan_ok(int log2n, const StageDesc *st12, int rnd_kind)
{
    const int cm = fastw64_multiplier_form(log2n, st12, rnd_kind);
    return cm == 1 || cm == 3; // (form 3 is never chosen in round mode)
}
const char *fastw64b_kernel_name(int direction) { return direction == 1 ? "k_ifft4096_w64" : "k_fft4096_w64"; }

hipError_t launch_fastw64b(int log2n, int direction, int rnd_kind, const StageDesc *st12, int in_cb, int dw, const void *in, void *out, const int2 *tw_all,
                           const int2 *h_tw, size_t nframes, hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    W64BArgs a;
    for (int s = 0; s < 12; ++s) a.st[s] = st12[s];
    a.in_cb = in_cb, a.dw = dw, a.native = native;
    if (native) return launch_fastw64_block_native(log2n, direction, rnd_kind, fastw64_multiplier_form(log2n, st12, rnd_kind), c, a, in, out, tw_all, nframes, stream);
    return launch_fastw64_block(log2n, direction, rnd_kind, fastw64_multiplier_form(log2n, st12, rnd_kind), c, a, in, out, tw_all, nframes, stream);
}

const char *fastw64_kernel_name(int direction) { return direction == 1 ? "k_ifft1024_w64" : "k_fft1024_w64"; }

hipError_t launch_fastw64(int log2n, int direction, int rnd_kind, const StageDesc *st10, int in_cb, int dw, const void *in, void *out, const int2 *tw_all, const int2 *h_tw,
                          size_t nframes, hipStream_t stream, int native)
{
    if (nframes == 0) return hipSuccess;
    UConsts c;
    for (int k = 0; k < 8; ++k) c.wr3[k] = h_tw[7 + k].x, c.wi3[k] = h_tw[7 + k].y;
    for (int k = 0; k < 4; ++k) c.wr2[k] = h_tw[3 + k].x, c.wi2[k] = h_tw[3 + k].y;
    W64Args a;
    for (int s = 0; s < 10; ++s) a.st[s] = st10[s];
    a.in_cb = in_cb, a.dw = dw, a.native = native;
    if (log2n < 10)
        return native ? launch_fastw64_short_native(log2n, direction, rnd_kind, c, a, in, out, tw_all, nframes, stream)
                      : launch_fastw64_short(log2n, direction, rnd_kind, c, a, in, out, tw_all, nframes, stream);
    const int cm = fastw64_multiplier_form(10, st10, rnd_kind);
    if (native) return launch_fastw64_native(direction, rnd_kind, cm, c, a, in, out, tw_all, nframes, stream); // the natural-order instances carry none of that code
#define INTFFT_W64N(R, CM)                                                                                                               \
    {                                                                                                                                   \
        if (direction == 1) launch_w64_kernel(k_ifft1024_w64<10, R, CM>, 10, c, a, in, out, tw_all, nframes, stream);                    \
        else launch_w64_kernel(k_fft1024_w64<10, R, CM>, 10, c, a, in, out, tw_all, nframes, stream);                                    \
    }
#define INTFFT_W64(R)                                                                                                                    \
    {                                                                                                                                   \
        if (cm == 1) INTFFT_W64N(R, 1) else if (cm == 3) INTFFT_W64N(R, 3) else INTFFT_W64N(R, 0)                                        \
    }
    if (rnd_kind == RND_TRUNC) INTFFT_W64(RND_TRUNC)
    else if (rnd_kind == RND_ROUND) INTFFT_W64(RND_ROUND)
    else INTFFT_W64(RND_UNSCALED)
#undef INTFFT_W64
#undef INTFFT_W64N
    return hipGetLastError();
}

} // namespace intfft
