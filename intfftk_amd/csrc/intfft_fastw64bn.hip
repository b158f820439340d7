// intfft_fastw64bn.hip -- the 64-bit block kernels (intfft_w64.hpp) at N = 2048 / 4096 in the cores' own beat orders (NAT instantiations: HALVES on the
// time side, BITREV on the frequency side), forward and inverse.  Its own translation unit for build time.
#include "intfft_w64.hpp"

namespace intfft {

hipError_t launch_fastw64_block_native(int log2n, int direction, int rnd_kind, int cm, const UConsts &c, const W64BArgs &a, const void *in, void *out,
                                       const int2 *tw_all, size_t nframes, hipStream_t stream)
{
#define INTFFT_W64B(LL, R, CM)                                                                                                           \
    {                                                                                                                                   \
        if (direction == 1) launch_w64b_kernel(k_ifft4096_w64<LL, R, CM, true>, LL, c, a, in, out, tw_all, nframes, stream);             \
        else launch_w64b_kernel(k_fft4096_w64<LL, R, CM, true>, LL, c, a, in, out, tw_all, nframes, stream);                             \
    }
#define INTFFT_W64BC(LL, R)                                                                                                              \
    {                                                                                                                                   \
        if (cm == 1) INTFFT_W64B(LL, R, 1) else INTFFT_W64B(LL, R, 3)                                                                    \
    }
#define INTFFT_W64BL(R)                                                                                                                  \
    {                                                                                                                                   \
        if (log2n == 11) INTFFT_W64BC(11, R) else INTFFT_W64BC(12, R)                                                                    \
    }
    if ((cm != 1 && cm != 3) || (rnd_kind == RND_ROUND && cm != 1)) return hipErrorInvalidValue; // (fastw64b_plan_ok)
    if (rnd_kind == RND_TRUNC) INTFFT_W64BL(RND_TRUNC)
    else if (rnd_kind == RND_ROUND) {
        if (log2n == 11) INTFFT_W64B(11, RND_ROUND, 1) else INTFFT_W64B(12, RND_ROUND, 1)
    } else INTFFT_W64BL(RND_UNSCALED)
#undef INTFFT_W64BL
#undef INTFFT_W64BC
#undef INTFFT_W64B
    return hipGetLastError();
}

} // namespace intfft
