// intfft_fast1024.hip -- packed-int16 wave kernel for N = 1024 (placeholder until the kernel lands).
#include "intfft_internal.hpp"

namespace intfft {

bool fast1024_supported(int, int, int, int, int, int, int, int, int) { return false; }

hipError_t launch_fast1024(const Fast1024Args &, const void *, void *, const int2 *, const unsigned *, size_t,
                           hipStream_t)
{
    return hipErrorNotSupported;
}

const char *fast1024_kernel_name() { return "k_fft1024_i16"; }

} // namespace intfft
