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


# ------------------------------------------------------------------------------------------------------------------------
# csrc/igemm_h2_dw.hip::conv_igemm_dw8u - the slice-unrolled k-loop of the 8-wave 256x256 convolution kernel.
# Restated: which 16-byte unit of the bordered fp16 operand / of the fp16 weight panel every LDS-DMA piece of every k-tile
# fetches, into which ring stage, in which iteration it is issued and waited for - and that every MFMA fragment read of k-tile t
# finds k-tile t's data (tap t % 9 of channel slice t // 9) in the stage it addresses, for every output row / column of the tile.
def _dw8u_schedule(nsl):
    """-> list of (iteration t, [('A'|'B', k-tile staged, ring stage)]) in issue order, incl. the prologue (t = -1)"""
    nt = 9 * nsl
    sched = [(-1, [("B", 0, 0), ("A", 0, 0), ("A", 1, 1), ("B", 1, 1)])]
    for t in range(nt):
        q = t % 9
        last = t // 9 == nsl - 1
        if last and q >= 7:
            sched.append((t, []))                        # tail iterations: nothing left to stage
        else:
            sched.append((t, [("B", t + 2, (q + 2) % 3), ("A", t + 2, (q + 2) % 3)]))
    return sched


@pytest.mark.parametrize("nsl", [1, 2, 3, 8])
def test_dw8u_ring_schedule_has_no_raw_or_war_hazard(nsl):
    """Ring of three stages, prefetch distance two, ONE barrier per k-tile (after the wave's own pieces of k-tile t+1 landed):
    every k-tile is staged exactly once, into stage (k-tile mod 3) - the compile-time stage of position q = t mod 9 because
    9 = 0 (mod 3) -; it is staged at least one barrier before its first read (RAW) and never while the k-tile it overwrites can
    still be read (WAR: reads of k-tile t happen in iterations t-1 (after the barrier) and t (before the barrier))."""
    nt = 9 * nsl
    staged_at, stage_of = {}, {}
    for t, pieces in _dw8u_schedule(nsl):
        for op, kt, st in pieces:
            assert (op, kt) not in staged_at, "staged twice"
            assert kt < nt and st == kt % 3
            staged_at[(op, kt)], stage_of[(op, kt)] = t, st
    for op in "AB":
        assert sorted(k for (o, k) in staged_at if o == op) == list(range(nt))
        for kt in range(nt):
            issue = staged_at[(op, kt)]
            # RAW: the first read of k-tile kt comes after the barrier of iteration kt - 1; the wait before that barrier
            # covers every piece issued up to iteration kt - 2 (pieces of iteration kt - 1 may still fly)
            assert issue <= kt - 2 or issue == -1, (op, kt, issue)
            # WAR: the stage held k-tile kt - 3, last read before the barrier of iteration kt - 3; the write is issued in
            # iteration kt - 2 (or the prologue)
            if kt >= 3:
                assert issue >= kt - 2


@pytest.mark.parametrize("shape", [(1, 32, 32, 32), (2, 16, 16, 64), (1, 64, 64, 96), (1, 8, 512, 64)], ids=str)
def test_dw8u_pieces_and_fragment_reads_agree(shape):
    B, H, W, C = shape
    Wp, HW, pad = W + 2, H * W, 1
    nsl = C // 32
    ATILE = 256 * 64
    total = B * (H + 2) * Wp * C
    toffx = []                                            # as the kernel builds it: byte offset of the k-tile staged at position q
    for q in range(9):
        tap = (q + 2) % 9
        ky, kx = divmod(tap, 3)
        toffx.append(((ky - pad) * Wp + (kx - pad)) * C * 2 + (64 if q + 2 >= 9 else 0))
    t0 = ((0 - pad) * Wp + (0 - pad)) * C * 2
    for tile_m in range(B * HW // 256):
        m0 = tile_m * 256
        lds = {}                                          # (stage, byte offset in stage) -> (k-tile, element offset fetched)
        actr_slice = 0                                    # + 64 bytes per finished slice

        def piece_a(off, stage, kt):
            for wave in range(8):
                for it in range(2):
                    for lane in range(64):
                        lrow = lane >> 2
                        ls = (lane & 3) ^ ((lrow >> 2) & 3)
                        m = m0 + wave * 32 + it * 16 + lrow
                        b, rem = divmod(m, HW)
                        oy, ox = divmod(rem, W)
                        src = ((b * (H + 2) + oy + 1) * Wp + ox + 1) * C * 2 + ls * 16 + actr_slice + off
                        assert src % 16 == 0 and 0 <= src // 2 and src // 2 + 8 <= total, "fetch outside the tensor"
                        lds[(stage, wave * 2048 + it * 1024 + lane * 16)] = (kt, src // 2)

        piece_a(t0, 0, 0)
        piece_a(t0 + C * 2, 1, 1)
        for t in range(9 * nsl):
            s, q = divmod(t, 9)
            # reads of k-tile t (set 1 in this iteration's first half; set 0 was read after the previous barrier): stage q % 3
            tap = t % 9
            ky, kx = divmod(tap, 3)
            for wr in range(4):
                for i in range(2):
                    for lr in range(32):
                        row = wr * 64 + i * 32 + lr
                        m = m0 + row
                        b, rem = divmod(m, HW)
                        oy, ox = divmod(rem, W)
                        for slot in range(4):             # k-slot s * 2 + lk: 8 channels each
                            so = (slot ^ ((lr >> 2) & 3)) << 4
                            kt, got = lds[(q % 3, row * 64 + so)]
                            want = ((b * (H + 2) + oy + ky) * Wp + ox + kx) * C + s * 32 + slot * 8
                            assert kt == t and got == want, (tile_m, t, row, slot)
            if not (s == nsl - 1 and q >= 7):
                piece_a(toffx[q], (q + 2) % 3, t + 2)
            if q == 8:
                actr_slice += 64
        assert ATILE == 8 * 2048
