// intfft_fastw64sn.hip -- the 64-bit wave kernels (intfft_w64.hpp) at N = 128 .. 512 in the cores' own beat orders (NAT instantiations, narrow
// multiplier form as intfft_fastw64s.hip).  Its own translation unit for build time.
#include "intfft_w64.hpp"

namespace intfft {

hipError_t launch_fastw64_short_native(int log2n, int direction, int rnd_kind, const UConsts &c, const W64Args &a, const void *in, void *out,
                                       const int2 *tw_all, size_t nframes, hipStream_t stream)
{
#define INTFFT_W64S(LL, R)                                                                                                               \
    {                                                                                                                                   \
        if (direction == 1) launch_w64_kernel(k_ifft1024_w64<LL, R, 1, true>, LL, c, a, in, out, tw_all, nframes, stream);               \
        else launch_w64_kernel(k_fft1024_w64<LL, R, 1, true>, LL, c, a, in, out, tw_all, nframes, stream);                               \
    }
#define INTFFT_W64SL(R)                                                                                                                  \
    {                                                                                                                                   \
        switch (log2n) {                                                                                                                \
        case 7: INTFFT_W64S(7, R) break;                                                                                                 \
        case 8: INTFFT_W64S(8, R) break;                                                                                                 \
        case 9: INTFFT_W64S(9, R) break;                                                                                                 \
        default: return hipErrorInvalidValue;                                                                                           \
        }                                                                                                                               \
    }
    if (rnd_kind == RND_TRUNC) INTFFT_W64SL(RND_TRUNC)
    else if (rnd_kind == RND_ROUND) INTFFT_W64SL(RND_ROUND)
    else INTFFT_W64SL(RND_UNSCALED)
#undef INTFFT_W64SL
#undef INTFFT_W64S
    return hipGetLastError();
}

} // namespace intfft
