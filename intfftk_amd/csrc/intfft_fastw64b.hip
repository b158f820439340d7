// intfft_fastw64b.hip -- the 64-bit block kernels (intfft_w64.hpp) at N = 2048 / 4096, forward: int_fftNk with results of 33 .. 64 bits
// (32-bit unscaled data: 43 / 44-bit results; wide scaled data) beyond what intfft_fast4096w.hip's 64-bit last round reaches.
// Multiplier forms 1 (narrow) and 3 (three-dword products); anything else stays on the generic kernel.
#include "intfft_w64.hpp"

namespace intfft {

hipError_t launch_fastw64_block(int log2n, int direction, int rnd_kind, int cm, const UConsts &c, const W64BArgs &a, const void *in, void *out,
                                const int2 *tw_all, size_t nframes, hipStream_t stream)
{
    if (direction == 1) return launch_fastw64_block_inv(log2n, rnd_kind, cm, c, a, in, out, tw_all, nframes, stream);
#define INTFFT_W64B(LL, R, CM) launch_w64b_kernel(k_fft4096_w64<LL, R, CM>, LL, c, a, in, out, tw_all, nframes, stream)
#define INTFFT_W64BC(LL, R)                                                                                                              \
    {                                                                                                                                   \
        if (cm == 1) INTFFT_W64B(LL, R, 1);                                                                                              \
        else INTFFT_W64B(LL, R, 3);                                                                                                      \
    }
#define INTFFT_W64BL(R)                                                                                                                  \
    {                                                                                                                                   \
        if (log2n == 11) INTFFT_W64BC(11, R) else INTFFT_W64BC(12, R)                                                                    \
    }
    if ((cm != 1 && cm != 3) || (rnd_kind == RND_ROUND && cm != 1)) return hipErrorInvalidValue; // (fastw64b_plan_ok)
    if (rnd_kind == RND_TRUNC) INTFFT_W64BL(RND_TRUNC)
    else if (rnd_kind == RND_ROUND) { // round mode with three-dword products spills 70-330 dwords: not instantiated
        if (log2n == 11) INTFFT_W64B(11, RND_ROUND, 1);
        else INTFFT_W64B(12, RND_ROUND, 1);
    } else INTFFT_W64BL(RND_UNSCALED)
#undef INTFFT_W64BL
#undef INTFFT_W64BC
#undef INTFFT_W64B
    return hipGetLastError();
}

} // namespace intfft
