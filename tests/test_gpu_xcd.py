"""The placement assumption of the half-line tiles (intfft_big2x.hip): workgroups b and b + 8 of a launch run on the same XCD
(the dispatcher hands consecutive workgroups to the eight XCDs in turn), so the two halves of a 128-byte line meet in ONE L2.  No result
depends on it -- only the HBM traffic of pass A / pass QA / the tiled 2-D column pass does (paired reads 1.05x, unpaired ~2x,
profiles/r04_reqbench.json) -- which is exactly why a box where it stops holding would go unnoticed by the parity tests.  This one fails loudly."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_blocks_b_and_b_plus_8_share_an_xcd():
    import torch

    from intfftk_amd.build import build_diag

    L = ctypes.CDLL(build_diag())
    L.diag_xcc_map.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    nblocks = 64 * 64  # one 64-frame chunk of N = 2^20: the grid of k_big2x_a
    out = torch.full((nblocks,), -1, dtype=torch.int32, device="cuda")
    for spin in (1, 200):  # a launch shorter and one longer than its own dispatch
        assert L.diag_xcc_map(out.data_ptr(), nblocks, spin, None) == 0
        torch.cuda.synchronize()
        x = out.cpu().numpy() & 15
        n_xcd = int(x.max()) + 1
        assert n_xcd in (1, 2, 4, 8), x[:32]
        assert np.array_equal(x[:-8], x[8:]), "blocks b and b + 8 landed on different XCDs: the half-line tiles lose their pairing on this box"
        if n_xcd == 8:  # MI355X in SPX mode: block b on XCD b mod 8
            assert len(set(x[:8].tolist())) == 8 and np.array_equal(x, np.resize(x[:8], nblocks)), x[:32]
