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
# csrc/igemm_h2_dw.hip::conv_igemm_dw - the k-loop of the 8-wave 256x256 convolution kernel with ASYMMETRIC staging (round 4): the
# older wave of every SIMD (waves 0-3) stages the rows of both waves of its SIMD, the younger (4-7) issues no LDS-DMA.
# Restated: which 16-byte unit of the bordered fp16 operand / of the fp16 weight panel every LDS-DMA piece of every k-tile fetches,
# into which ring stage, in which iteration and at which point of it (IN segment A, or LATE = after the vmcnt wait, in front of the
# barrier) it is issued, what the counted vmcnt wait of every iteration therefore covers - and that every MFMA fragment read of
# k-tile t finds k-tile t's data in the stage it addresses, for every output row of the tile, incl. 1x1 K-segments.
def _dw_schedule(nt):
    """-> list of (iteration t, when, [(operand, piece it, k-tile staged, ring stage)]) in ISSUE ORDER of a staging wave, incl. the
    prologue (t = -1); piece it: 0, 1 = own rows, 2, 3 = the partner's"""
    sched = [(-1, "prologue", [("B", it, 0, 0) for it in range(4)] + [("A", it, 0, 0) for it in range(4)] +
              [("A", it, 1, 1) for it in range(4)] + [("B", it, 1, 1) for it in range(4)])]
    for t in range(nt - 2):                               # steady iterations: k-tile t + 2 exists
        st = (t + 2) % 3
        sched.append((t, "in A", [("B", 0, t + 2, st), ("B", 1, t + 2, st), ("A", 0, t + 2, st), ("A", 1, t + 2, st)]))
        sched.append((t, "late", [("B", 2, t + 2, st), ("B", 3, t + 2, st), ("A", 2, t + 2, st), ("A", 3, t + 2, st)]))
    return sched


@pytest.mark.parametrize("nt", [4, 9, 18, 72, 75])
def test_dw_ring_schedule_has_no_raw_or_war_hazard(nt):
    """Rings of three stages, prefetch distance two, ONE barrier per k-tile.  In iteration t the staging wave issues its own four
    pieces of k-tile t+2 inside segment A, then waits with vmcnt(4) - in issue order that leaves exactly those four in flight, so the
    wait covers the LATE pieces of the previous iteration too -, then issues the partner's four pieces of k-tile t+2, then meets the
    barrier.  Every piece of k-tile t+1 has therefore landed before barrier(t), after which its fragments are first read (RAW); a
    stage is rewritten in iteration t only after every read of the k-tile it held (t-1: read in iterations t-2 after the barrier
    and t-1 before it) lies behind barrier(t-1), which the issuing wave has passed (WAR)."""
    queue, landed_by_wait = [], {}                        # issue-ordered (k-tile, operand, piece); per iteration: what its wait retires
    staged = {}
    for t, when, pieces in _dw_schedule(nt):
        if when == "late":                                # the wait of iteration t sits between "in A" and "late"
            keep = queue[-4:]                             # s_waitcnt vmcnt(4): the four youngest may still fly
            assert all(kt == t + 2 for kt, _, _ in keep)
            landed_by_wait[t] = {q for q in queue[:-4]}
            queue = keep
        for op, it, kt, st in pieces:
            assert (op, it, kt) not in staged, "staged twice"
            assert kt < nt and st == kt % 3
            staged[(op, it, kt)] = (t, when)
            queue.append((kt, op, it))
        if when == "prologue":                            # vmcnt(8): k-tile 0 landed
            assert {kt for kt, _, _ in queue[:-8]} == {0} and len(queue[:-8]) == 8
            queue = queue[-8:]
    for op in "AB":
        for it in range(4):
            assert sorted(k for (o, i, k) in staged if o == op and i == it) == list(range(nt))
    done = set()
    for t in range(nt - 2):
        done |= landed_by_wait[t]
        # RAW: before barrier(t) every piece of k-tile t + 1 has landed
        assert all((t + 1, op, it) in done or t + 1 <= 0 for op in "AB" for it in range(4)), t
    for (op, it, kt), (t, when) in staged.items():
        if kt >= 3:                                       # WAR: stage kt % 3 held k-tile kt - 3, whose last reads precede barrier(kt - 3)
            assert t == kt - 2                            # ... and this piece is issued in iteration kt - 2 > kt - 3


@pytest.mark.parametrize("shape", [(1, 32, 32, 32, 0, 0), (2, 16, 16, 64, 0, 0), (1, 4, 512, 32, 0, 0), (1, 10, 128, 32, 0, 0),
                                   (1, 16, 16, 32, 64, 32), (1, 16, 32, 64, 32, 0)], ids=str)
def test_dw_pieces_and_fragment_reads_agree(shape):
    """Piece -> (rows, LDS destination, source address) of the staging waves against the fragment reads of all eight waves, for the
    3x3 part and the 1x1 K-segments that follow it (plain fp16 NHWC tensors, no border)."""
    B, H, W, C, C1, C2 = shape
    Wp, HW, pad = W + 2, H * W, 1
    nt_main = 9 * (C // 32)
    nt = nt_main + (C1 + C2) // 32
    total = B * (H + 2) * Wp * C
    for tile_m in range(B * HW // 256):
        m0 = tile_m * 256
        lds = {}                                          # (stage, byte offset in stage) -> (k-tile, tensor, element offset fetched)

        def piece_a(kt):
            """the four activation pieces of k-tile kt as the staging waves (0-3) issue them"""
            stage = kt % 3
            for wave in range(4):
                for it in range(4):
                    rows0 = (wave + (it >> 1) * 4) * 32 + (it & 1) * 16
                    for lane in range(64):
                        lrow = lane >> 2
                        ls = (lane & 3) ^ ((lrow >> 2) & 3)
                        m = m0 + rows0 + lrow
                        if kt < nt_main:
                            c, tap = divmod(kt, 9)
                            ky, kx = divmod(tap, 3)
                            b, rem = divmod(m, HW)
                            oy, ox = divmod(rem, W)
                            src = ((b * (H + 2) + oy + 1) * Wp + ox + 1) * C * 2 + ls * 16 + ((ky - pad) * Wp + (kx - pad)) * C * 2 + c * 64
                            assert src % 16 == 0 and 0 <= src // 2 and src // 2 + 8 <= total, "fetch outside the tensor"
                            what = ("x", src // 2)
                        else:
                            j = kt - nt_main
                            seg, cs, c = (1, C1, j) if j < C1 // 32 else (2, C2, j - C1 // 32)
                            src = m * cs * 2 + ls * 16 + c * 64
                            assert src // 2 + 8 <= B * HW * cs
                            what = (f"seg{seg}", src // 2)
                        lds[(stage, rows0 * 64 + lane * 16)] = (kt,) + what

        piece_a(0)
        piece_a(1)
        for t in range(nt):
            for wr in range(4):                           # fragment reads of k-tile t: stage t % 3, every wave's 64 rows
                for i in range(2):
                    for lr in range(32):
                        row = wr * 64 + i * 32 + lr
                        m = m0 + row
                        b, rem = divmod(m, HW)
                        oy, ox = divmod(rem, W)
                        for slot in range(4):             # k-slot s * 2 + lk: 8 channels each
                            so = (slot ^ ((lr >> 2) & 3)) << 4
                            kt, tensor, got = lds[(t % 3, row * 64 + so)]
                            if t < nt_main:
                                c, tap = divmod(t, 9)
                                ky, kx = divmod(tap, 3)
                                want = ("x", ((b * (H + 2) + oy + ky) * Wp + ox + kx) * C + c * 32 + slot * 8)
                            else:
                                j = t - nt_main
                                seg, cs, c = (1, C1, j) if j < C1 // 32 else (2, C2, j - C1 // 32)
                                want = (f"seg{seg}", m * cs + c * 32 + slot * 8)
                            assert kt == t and (tensor, got) == want, (tile_m, t, row, slot)
            if t + 2 < nt:
                piece_a(t + 2)
