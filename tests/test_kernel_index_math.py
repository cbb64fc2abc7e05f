"""CPU restatement of the operand staging of csrc/igemm_h2_nn.hip (the x-halo run), element by element: which 16-byte unit of the
zero-bordered fp16 operand every LDS-DMA lane fetches, where it lands, and which unit every MFMA fragment read picks up for every
output pixel, tap and k-slot.  An executable statement of the layout the kernel's comments describe (row r * (W + 2) + xl + kx of
the run, XOR swizzle keyed by the LDS row, pieces beyond the run clamped to its last row): an edit of the address arithmetic in the
kernel has to keep this file true.  No GPU."""
import numpy as np
import pytest

APW = 5                     # run pieces per wave (igemm_h2_nn.hip)


def stage_run(B, H, W, C, m0, c, ky):
    """-> lds[unit] = element offset (in fp16 elements of the bordered [B, H+2, W+2, C] tensor) of the 8 halves the unit holds,
    for the run of (slice c, ky) of the tile starting at output pixel m0; unit = LDS row * 4 + physical slot"""
    Wp, HW = W + 2, H * W
    L = min(W, 256)
    Lp = L + 2
    NR = (256 // L) * Lp
    b, rem = divmod(m0, HW)
    oy0, ox0 = divmod(rem, W)
    xrun = ((b * (H + 2) + oy0) * Wp + ox0) * C          # element offset of the run of ky = 0, slice 0
    total = B * (H + 2) * Wp * C
    lds = -np.ones(4 * APW * 16 * 4, dtype=np.int64)
    for wave in range(4):
        for j in range(APW):
            for lane in range(64):
                lrow = lane >> 2
                ls = (lane & 3) ^ ((lrow >> 2) & 3)
                prow = min((wave + 4 * j) * 16 + lrow, NR - 1)
                src_bytes = xrun * 2 + ky * Wp * C * 2 + c * 64 + prow * C * 2 + ls * 16
                assert src_bytes % 16 == 0 and src_bytes // 2 + 8 <= total, "fetch outside the tensor"
                lds[((wave + 4 * j) * 1024 + lane * 16) // 16] = src_bytes // 2
    return lds


@pytest.mark.parametrize("shape", [(1, 32, 32, 64), (2, 64, 64, 32), (1, 128, 128, 32), (1, 256, 256, 32), (1, 8, 512, 32)], ids=str)
def test_x_halo_run_serves_the_three_kx_taps(shape):
    B, H, W, C = shape
    Wp, HW = W + 2, H * W
    L = min(W, 256)
    Lp, lsh = L + 2, min(W, 256).bit_length() - 1
    for tile in range(B * HW // 256):
        m0 = tile * 256
        for c in range(C // 32):
            for ky in range(3):
                lds = stage_run(B, H, W, C, m0, c, ky)
                for wave in range(4):
                    for i in range(2):
                        for lr in range(32):
                            ml = wave * 64 + i * 32 + lr
                            row0 = (ml >> lsh) * Lp + (ml & (L - 1))
                            m = m0 + ml
                            b, r2 = divmod(m, HW)
                            oy, ox = divmod(r2, W)
                            for kx in range(3):
                                row = row0 + kx
                                for slot in range(4):       # k-slot s * 2 + lk: 8 channels each
                                    got = lds[(row * 64 + ((slot ^ ((row >> 2) & 3)) << 4)) // 16]
                                    want = ((b * (H + 2) + oy + ky) * Wp + ox + kx) * C + c * 32 + slot * 8
                                    assert got == want, (tile, c, ky, ml, kx, slot)


def test_swizzle_is_conflict_free_for_shifted_rows():
    """ds_read_b128 serves 16 lanes per LDS cycle ({0-3, 12-15, 20-27} and so on): with 64-byte rows and the key (row >> 2) & 3, the 16
    rows such a group reads - consecutive rows in any alignment, i.e. also the rows shifted by kx - hit 16 distinct 16-byte bank slots."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for shift in range(0, 70):
        for slot in range(4):
            for g in groups:
                banks = {(((lr + shift) * 64 + ((slot ^ (((lr + shift) >> 2) & 3)) << 4)) // 16) % 16 for lr in g}
                assert len(banks) == 16, (shift, slot)
