"""Command-line front end mirroring the reference testbenches' flow (text files in, text files out).

    python -m intfftk_amd.cli single --nfft 7 --mode TRUNCATE di_single.dat dout_single.dat
    python -m intfftk_amd.cli pair   --nfft 7 --mode UNSCALED di_double.dat dout_pair.dat [--reference-wiring]

`single` = fft_signle_test.vhd (int_fft_single_path, natural in -> natural out, one (re, im) per line);
`pair`   = fft_double_test.vhd (int_fft_ifft_pair, four integers per beat in, top 17 bits per beat out).
Mode names are the testbench's set_mode() strings (fft_signle_test.vhd:80-112).  Needs a HIP device.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="intfftk_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("flow", choices=["single", "pair"])
    ap.add_argument("infile")
    ap.add_argument("outfile")
    ap.add_argument("--nfft", type=int, default=7, help="NFFT generic = log2 of the length")
    ap.add_argument("--mode", default="UNSCALED", choices=["UNSCALED", "ROUNDING", "TRUNCATE"])
    ap.add_argument("--data-width", type=int, default=16)
    ap.add_argument("--twdl-width", type=int, default=16)
    ap.add_argument("--xseries", default="NEW", choices=["NEW", "OLD"])
    ap.add_argument("--reference-wiring", action="store_true",
                    help="pair: reproduce the Q0_IM/Q1_RE slice mix-up of int_fft_ifft_pair.vhd:332-335")
    a = ap.parse_args(argv)

    import torch

    from . import int_fft_ifft_pair, int_fft_single_path, set_mode, textio

    fmt, rnd = set_mode(a.mode)
    n = 1 << a.nfft
    if a.flow == "single":
        x = textio.read_di_single(a.infile, n)
        core = int_fft_single_path(a.nfft, a.data_width, a.twdl_width, fmt, rnd, a.xseries)
    else:
        x = textio.read_di_double(a.infile, n)
        core = int_fft_ifft_pair(a.nfft, a.data_width, a.twdl_width, fmt, rnd, a.xseries)
    dt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
    y = core(torch.from_numpy(x.astype(dt)).cuda()).cpu().numpy()
    if core.out_container == 16:  # results beyond 64 bits: Python integers (the text formats are width-agnostic)
        from .engine import wide_to_int

        y = wide_to_int(y)
    if a.flow == "single":
        textio.write_di_single(a.outfile, y)
    else:
        textio.write_dout_pair(a.outfile, y, core.out_bits, a.reference_wiring)
    print("%s: %d frame(s) of %d points, %s, %d -> %d bits, kernel %s" %
          (a.flow, x.shape[0], n, a.mode, core.in_bits, core.out_bits, core.info["kernel_name"]))
    core.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
