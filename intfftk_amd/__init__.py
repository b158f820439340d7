"""intfftk_amd -- MI355X-native fixed-point radix-2 FFT/IFFT engine (drop-in for intfftk's int_fftNk /
int_ifftNk hot path).  The product is libintfft.so (HIP kernels + C-ABI, include/intfft.h); this
package is its thin host-side mirror of the reference's entity interface."""
from ._capi import (ERR_INVALID, ERR_NO_DEVICE, ERR_NULL, ERR_UNSUPPORTED, FWD, INV, LIB_PATH, OK, PAIR,
                    IntFFTError)
from .engine import (FrameStream, IntFFTCore, exec_sharded, int_fft_2d, int_fft_ifft_pair, int_fft_single_path, int_fftNk, int_ifftNk,
                     set_mode)

__all__ = ["IntFFTCore", "FrameStream", "int_fftNk", "int_ifftNk", "int_fft_single_path", "int_fft_ifft_pair", "int_fft_2d", "set_mode", "exec_sharded",
           "IntFFTError", "LIB_PATH", "OK", "FWD", "INV", "PAIR", "ERR_INVALID", "ERR_UNSUPPORTED", "ERR_NULL",
           "ERR_NO_DEVICE"]
