// intfft_fastw64n.hip -- the 64-bit wave kernels (intfft_w64.hpp) at N = 1024 in the cores' own beat orders (NAT instantiations: HALVES on the
// time side, BITREV on the frequency side; int_fftNk.vhd:60-75 / int_ifftNk.vhd:60-75 without the input buffer / int_bitrev_order blocks).
// Its own translation unit for build time (as many instances as intfft_fastw64.hip).
#include "intfft_w64.hpp"

namespace intfft {

hipError_t launch_fastw64_native(int direction, int rnd_kind, int cm, const UConsts &c, const W64Args &a, const void *in, void *out, const int2 *tw_all,
                                 size_t nframes, hipStream_t stream)
{
#define INTFFT_W64N(R, CM)                                                                                                               \
    {                                                                                                                                   \
        if (direction == 1) launch_w64_kernel(k_ifft1024_w64<10, R, CM, true>, 10, c, a, in, out, tw_all, nframes, stream);              \
        else launch_w64_kernel(k_fft1024_w64<10, R, CM, true>, 10, c, a, in, out, tw_all, nframes, stream);                              \
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
