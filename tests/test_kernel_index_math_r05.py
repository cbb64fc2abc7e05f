"""CPU restatement of the index arithmetic of round 5's kernels, element by element - an edit of the address arithmetic in the kernels has to keep
this file true.  No GPU.

  csrc/igemm_h2_dh.hip   (conv_igemm_dh<BMT>: 128x256 / 256x128 tiles, four waves): which 16-byte unit of the zero-bordered fp16 operand and of the
                         weight panel every LDS-DMA lane fetches and where it lands; which unit every MFMA fragment read picks up; that the
                         pieces of four waves cover both operand tiles exactly once; the counted vmcnt (RAW) and the ring reuse (WAR) of the
                         k-loop; the split-K cursor of a part that starts mid-way (tap fastest, then the slice, then the 1x1 segments)
  csrc/gemm_h16.hip      (gemm_strided_h16): the LDS image of both staging forms (eight values per 16-byte store / two-byte scatter), the
                         fragment reads, and bank-conflict freedom of a ds_read_b128 under the guide's lane groups
"""
import numpy as np
import pytest

KEY = lambda r: (r >> 2) & 3          # XOR key of an LDS row: 16-byte slot q of row r sits at physical slot q ^ KEY(r)


# ---- conv_igemm_dh -------------------------------------------------------------------------------------------------------------------------------
def dh_stage_tile(BMT, rows_src):
    """One operand tile of a k-tile as the kernel's four waves stage it.  rows_src(row) -> element offset of that row's 32 halves (the k-tile's
    slice) in its source tensor.  -> lds[row * 4 + physical slot] = element offset of the 8 halves the unit holds."""
    rows = BMT
    np_ = rows // 64                      # 16-row pieces per wave
    lds = -np.ones(rows * 4, dtype=np.int64)
    for wave in range(4):
        for it in range(np_):
            base_row = wave * (rows // 4) + it * 16
            for lane in range(64):
                lrow = lane >> 2
                ls = (lane & 3) ^ KEY(lrow)                     # logical slot this lane FETCHES
                src = rows_src(base_row + lrow) + ls * 8
                dst_byte = base_row * 64 + lane * 16            # an LDS-DMA instruction writes lane l's 16 bytes at base + 16 l
                assert lds[dst_byte // 16] == -1, "two lanes land on one unit"
                lds[dst_byte // 16] = src
    assert (lds >= 0).all(), "a unit of the tile is never staged"
    return lds


@pytest.mark.parametrize("BMT", [128, 256])
def test_dh_staging_and_fragment_reads(BMT):
    BNT = 384 - BMT
    B, H, W, C, N = 2, 16, 16, 64, BNT
    Wp, HW, K = W + 2, H * W, 9 * 64
    m0, n0 = BMT, 0                        # the second row tile
    for c in range(C // 32):
        for tap in range(9):
            ky, kx = divmod(tap, 3)
            assert ky == (tap * 11) >> 5   # the kernel's division-free tap / 3

            def act_row(r):                # centre pixel of output row m0 + r, shifted by the tap, slice c (fp16 elements of the bordered tensor)
                m = m0 + r
                b, rem = divmod(m, HW)
                oy, ox = divmod(rem, W)
                return ((b * (H + 2) + oy + 1 + ky - 1) * Wp + ox + 1 + kx - 1) * C + c * 32

            kt = c * 9 + tap
            la = dh_stage_tile(BMT, act_row)
            # weights: block layout of the fp16 panels (ops.order_conv_weight_w16): [n / 32][k / 8][32 rows][8] - the staging pointer of a lane is
            # (n >> 5) K 64 + (n & 31) 16 + ls 512 bytes, advanced by 2048 bytes per k-tile (a slot is 512 bytes away, not 16)
            lb = -np.ones(BNT * 4, dtype=np.int64)
            for wave in range(4):
                for it in range(BNT // 64):
                    for lane in range(64):
                        lrow = lane >> 2
                        ls = (lane & 3) ^ KEY(lrow)
                        n = n0 + wave * (BNT // 4) + it * 16 + lrow
                        src_bytes = (n >> 5) * K * 64 + (n & 31) * 16 + ls * 512 + kt * 2048
                        lb[((wave * (BNT // 4) + it * 16) * 64 + lane * 16) // 16] = src_bytes // 2
            assert (lb >= 0).all()
            # fragment reads: wave (wr, wc), MFMA tile i / j, lane (lr, lk), k16 step s -> 8 halves = channels (s * 2 + lk) * 8 .. + 7 of the k-tile
            for wave in range(4):
                wr, wc = (wave >> 1, wave & 1) if BMT == 128 else (wave, 0)
                for lr in range(32):
                    for lk in range(2):
                        for s_ in range(2):
                            soff = ((s_ * 2 + lk) ^ KEY(lr)) << 4
                            for i in range(2):
                                row = wr * 64 + i * 32 + lr
                                got = la[(row * 64 + soff) // 16]
                                assert got == act_row(row) + (s_ * 2 + lk) * 8, (wave, lr, lk, s_, i)
                            for j in range(4):
                                row = wc * 128 + j * 32 + lr
                                n = n0 + row
                                got = lb[(row * 64 + soff) // 16]
                                # element (n, k) of the panel [N32][K] in block order: block n >> 5, octet k >> 3, row n & 31
                                k = kt * 32 + (s_ * 2 + lk) * 8
                                want = (n >> 5) * K * 32 + (k >> 3) * 256 + (n & 31) * 8
                                assert got == want, (wave, lr, lk, s_, j)


def test_dh_vmcnt_and_ring_reuse():
    """Issue order of one wave: prologue B(0) A(0) A(1) B(1), then per iteration t the six pieces of k-tile t + 2; `s_waitcnt vmcnt(6)` retires, in
    issue order, everything but the six youngest pieces.  RAW: at the barrier of iteration t (and at the prologue's) every piece of k-tile t + 1 (0)
    has landed.  WAR: the stage written in iteration t, (t + 2) % 3, held k-tile t - 1, whose last reads precede the barrier of iteration t - 1."""
    for NPA, NPB in ((2, 4), (4, 2)):
        for nt in (4, 5, 9, 72):
            issued = []                                     # k-tile of every piece in issue order
            issued += [0] * NPB + [0] * NPA + [1] * NPA + [1] * NPB
            landed = lambda: set(issued[:max(0, len(issued) - 6)])          # after vmcnt(6)
            assert 0 in landed() and issued[-6:].count(0) == 0
            for t in range(nt - 2):
                stage_written = (t + 2) % 3
                assert stage_written == (t - 1) % 3         # the ring stage of k-tile t - 1: read before the barrier of iteration t - 1
                issued += [t + 2] * (NPA + NPB)
                done = issued[:len(issued) - 6]
                assert done.count(t + 1) == NPA + NPB, (NPA, nt, t)         # all of k-tile t + 1 retired before its fragments are read
            # tail: vmcnt(0)


@pytest.mark.parametrize("segs", [(0, 0), (64, 0), (64, 32)])
def test_dh_split_k_cursor(segs):
    """A split-K part starts at k-tile t0 = floor(y * ntot / S): the cursor (segment, slice, tap) the kernel computes there equals the cursor a
    part that walked from k-tile 0 has after t0 steps."""
    C, taps = 96, 9
    c1, c2 = segs
    conv_tiles = taps * (C // 32)
    ntot = conv_tiles + c1 // 32 + c2 // 32

    def walk(n):
        seg, slc, tap, seg_slices = 0, 0, 0, C // 32
        for _ in range(n):
            if slc == seg_slices:                            # pieceA(it == 0): this segment is staged, on to the next tensor
                seg += 1
                seg_slices = (c1 if seg == 1 else c2) // 32
                slc, tap = 0, 0
            # stage one k-tile
            if seg != 0:
                slc += 1
            else:
                tap += 1
                if tap == taps:
                    tap, slc = 0, slc + 1
        if slc == seg_slices and n < ntot:                   # normalise: the switch happens lazily, at the next staging
            seg += 1
            seg_slices = (c1 if seg == 1 else c2) // 32
            slc, tap = 0, 0
        return seg, slc, tap

    for S in (2, 4):
        for y in range(S):
            t0 = (y * ntot) // S
            if t0 < conv_tiles:
                cur = (0, t0 // taps, t0 % taps)
            else:
                rem = t0 - conv_tiles
                cur = (1, rem, 0) if rem < c1 // 32 else (2, rem - c1 // 32, 0)
            if t0 < ntot:
                assert cur == walk(t0), (S, y, t0, cur, walk(t0))


# ---- gemm_strided_h16 ----------------------------------------------------------------------------------------------------------------------------
def test_gemm_h16_lds_image_and_fragment_reads():
    NT = 256
    # KCONTIG form: unit = (row, 8 consecutive k)
    img = -np.ones((128 * 64 // 2,), dtype=np.int64)        # per fp16 element of the tile: row * 32 + k
    for tid in range(NT):
        for it in range(2):
            row, q = (tid >> 2) + it * 64, tid & 3
            d = row * 64 + ((q ^ KEY(row)) << 4)
            for j in range(8):
                assert img[d // 2 + j] == -1
                img[d // 2 + j] = row * 32 + q * 8 + j
    assert (img >= 0).all()
    # transposed form: four rows at one k per thread and iteration, two-byte scatter: the SAME image
    img2 = -np.ones_like(img)
    for tid in range(NT):
        for it in range(4):
            idx = tid + it * NT
            kr, rq = idx >> 5, idx & 31
            for j in range(4):
                row = rq * 4 + j
                d = row * 64 + (((kr >> 3) ^ KEY(row)) << 4) + (kr & 7) * 2
                assert img2[d // 2] == -1
                img2[d // 2] = row * 32 + kr
    assert np.array_equal(img, img2)
    # fragment reads of the 32x32x16 MFMA: lane (lr, lk), k16 step s -> k = (s * 2 + lk) * 8 .. + 7 of row base + lr
    for wr in range(2):
        for i in range(2):
            for lr in range(32):
                for lk in range(2):
                    for s_ in range(2):
                        row = wr * 64 + i * 32 + lr
                        off = row * 64 + (((s_ * 2 + lk) ^ KEY(lr)) << 4)
                        got = img[off // 2:off // 2 + 8]
                        assert list(got) == [row * 32 + (s_ * 2 + lk) * 8 + j for j in range(8)]
    # ds_read_b128 lane groups of the guide (MI355X_MICROARCH.md, LDS): {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: 16 lanes x 4 dwords must hit
    # 64 distinct banks (bank = (byte address / 4) mod 64)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for s_ in range(2):
        for g in groups:
            banks = []
            for lane in g:
                lr, lk = lane & 31, lane >> 5
                off = lr * 64 + (((s_ * 2 + lk) ^ KEY(lr)) << 4)
                banks += [((off // 4) + w) % 64 for w in range(4)]
            assert len(set(banks)) == 64, (s_, g)


# ---- round 6: the 512 x 128 form of the 8-wave kernel (csrc/igemm_h2_dw.hip, SHAPE 1) ---------------------------------------------------------
def _dw_rows_of_piece(wave, it, shape):
    """rows of activation piece `it` of staging wave `wave` (rows_of_piece): BMT / 8 rows per wave, NAO own pieces then the partner's"""
    bmt = 512 if shape else 256
    nao = bmt // 128
    return (wave + (it // nao) * 4) * (bmt // 8) + (it % nao) * 16


def _dw_rows_of_piece_b(wave, it, shape):
    bnt = 128 if shape else 256
    nbo = bnt // 128
    return (wave + (it // nbo) * 4) * (bnt // 8) + (it % nbo) * 16


@pytest.mark.parametrize("shape", [0, 1], ids=["256x256", "512x128"])
def test_dw_both_tile_shapes_stage_every_row_exactly_once(shape):
    """The four older waves (0-3) stage the rows of both waves of their SIMD: every activation row of the tile (256 | 512) and every weight
    row (256 | 128) lands exactly once per k-tile, in 16-row pieces, own pieces first (they are issued inside segment A), the partner's
    after the vmcnt wait; the LDS destination of a piece is its rows x 64 bytes, so the tile image is dense."""
    bmt, bnt = (512, 128) if shape else (256, 256)
    nao, nbo = bmt // 128, bnt // 128
    seen_a, seen_b = [], []
    for wave in range(4):
        for it in range(2 * nao):
            r0 = _dw_rows_of_piece(wave, it, shape)
            owner = r0 // (bmt // 8)
            assert owner == (wave if it < nao else wave + 4)          # own rows first, then the partner's (wave + 4: the same SIMD)
            seen_a += list(range(r0, r0 + 16))
        for it in range(2 * nbo):
            r0 = _dw_rows_of_piece_b(wave, it, shape)
            assert r0 // (bnt // 8) == (wave if it < nbo else wave + 4)
            seen_b += list(range(r0, r0 + 16))
    assert sorted(seen_a) == list(range(bmt)) and sorted(seen_b) == list(range(bnt))
    # LDS budget: three stages per operand inside the 160 KB of a CU, under the epilogue's 136 KB landing zone
    assert 3 * bmt * 64 + 3 * bnt * 64 <= 128 * 1088 <= 160 * 1024


def test_dw_512x128_fragment_rows_and_epilogue_coordinates():
    """SHAPE 1 stacks the eight waves along the pixels: wave w reads A fragments of rows [64 w, 64 w + 64) (two 32-row MFMA tiles) and the B
    fragments of all 128 columns (four 32-column tiles) - the 64 x 128 wave tile of the square form, so the epilogue is called with
    row0 = m0 + 64 w, colw = n0 and record index tile_m * 8 + w: the eight waves tile the 512 x 128 output exactly once and the
    64-row column records of a tile are consecutive."""
    cover = set()
    recs = []
    for wave in range(8):
        wr, wc = wave, 0
        for i in range(2):
            for lr in range(32):
                arow = (wr * 64 + lr) * 64 + i * 32 * 64
                assert arow // 64 == wave * 64 + i * 32 + lr < 512
        for j in range(4):
            for lr in range(32):
                brow = (wc * 128 + lr) * 64 + j * 32 * 64
                assert brow // 64 == j * 32 + lr < 128
        for r in range(64):
            for c in range(128):
                cover.add((wr * 64 + r, wc * 128 + c))
        recs.append(0 * 8 + wr)                                         # tile_m * (BMT / 64) + wr at tile_m = 0
    assert len(cover) == 512 * 128 and recs == list(range(8))


@pytest.mark.parametrize("nt", [4, 9, 36, 108])
def test_dw_512x128_vmcnt_schedule(nt):
    """Issue order of a staging wave in SHAPE 1 and what its counted waits retire.  Prologue: B(0) [2 pieces], A(0) [8], A(1) [8], B(1) [2],
    then vmcnt(10) -> k-tile 0 has landed.  Iteration t: own pieces of k-tile t+2 inside segment A (1 weight + 4 activation), vmcnt(5) -
    everything older has landed, in particular the partner's pieces of k-tile t+1 issued late in iteration t-1 -, then the partner's
    1 + 4 pieces of k-tile t+2, then the barrier.  RAW: every piece of k-tile t+1 is retired before barrier(t)."""
    NAO, NBO = 4, 1
    queue = []
    landed = set()

    def issue(op, kt, pieces):
        for it in pieces:
            queue.append((op, kt, it))

    def wait(n):
        nonlocal queue
        for q in queue[:len(queue) - n] if n else queue:
            landed.add(q)
        queue = queue[len(queue) - n:] if n else []

    issue("B", 0, range(2 * NBO)); issue("A", 0, range(2 * NAO)); issue("A", 1, range(2 * NAO)); issue("B", 1, range(2 * NBO))
    wait(2 * NAO + 2 * NBO)
    assert all(("A", 0, it) in landed for it in range(2 * NAO)) and all(("B", 0, it) in landed for it in range(2 * NBO))
    for t in range(nt - 2):
        issue("B", t + 2, range(NBO)); issue("A", t + 2, range(NAO))
        wait(NBO + NAO)
        issue("B", t + 2, range(NBO, 2 * NBO)); issue("A", t + 2, range(NAO, 2 * NAO))
        # barrier(t): k-tile t + 1 complete
        assert all(("A", t + 1, it) in landed for it in range(2 * NAO)) and all(("B", t + 1, it) in landed for it in range(2 * NBO)), t
    wait(0)                                                             # the tail waits with vmcnt(0)
    for kt in range(nt):
        assert all(("A", kt, it) in landed for it in range(2 * NAO)) and all(("B", kt, it) in landed for it in range(2 * NBO))


# ---- round 6: the lean apply pass of the GroupNorm backward (csrc/norm_bwd.hip gn_bwd_apply_lean_kernel) ------------------------------------
def _lean_grid(B, H, W, C8):
    """the launcher's geometry: at most 4096 workgroups of 256 threads, a thread count that is a multiple of C8"""
    items = B * H * W * C8
    wg = min(4096, (items + 255) // 256)
    while wg > 0 and (wg * 256) % C8 != 0:
        wg -= 1
    if wg == 0:
        return None
    PS = wg * 256 // C8
    HW = H * W
    dB, remp = divmod(PS, HW)
    dY, dX = divmod(remp, W)
    return wg, PS, dB, dY, dX


@pytest.mark.parametrize("shape", [(2, 32, 32, 32), (3, 6, 10, 8), (5, 8, 8, 96), (1, 64, 64, 16), (4, 4, 4, 128), (7, 5, 3, 24), (2, 256, 256, 32)], ids=str)
def test_gn_bwd_lean_apply_pixel_walk_and_border(shape):
    """A thread of the lean apply pass keeps channel octet gid % C8 and walks pixels pl, pl + PS, ... of the B*H*W pixels, carrying
    (sample, y, x) forward by the decomposition of PS = dB * HW + dY * W + dX with two carries - no division in the loop.  Restated: the
    walk visits every (pixel, octet) pair exactly once, its (sample, y, x) is the pixel's own at every step, the sample changes
    monotonically (the per-sample constants are re-loaded only then), and the second loop writes every frame pixel of the zero-bordered
    operand exactly once per octet and no interior pixel."""
    B, H, W, C8 = shape
    g = _lean_grid(B, H, W, C8)
    assert g is not None
    wg, PS, dB, dY, dX = g
    assert (wg * 256) % C8 == 0 and PS * C8 == wg * 256 and dB * H * W + dY * W + dX == PS and 0 <= dX < W and 0 <= dY < H
    total, HW = B * H * W, H * W
    lanes = range(PS) if PS <= 4096 else list(range(0, PS, max(1, PS // 512))) + [PS - 1]     # every pixel lane, or a sample of them on big grids
    seen = 0
    for pl in lanes:
        b, rem = divmod(pl, HW)
        y, x = divmod(rem, W)
        last_b = -1
        px = pl
        while px < total:
            assert (b, y, x) == (px // HW, (px % HW) // W, px % W), (pl, px)
            assert b >= last_b
            last_b = b
            seen += 1
            x += dX
            if x >= W:
                x -= W
                y += 1
            y += dY
            if y >= H:
                y -= H
                b += 1
            b += dB
            px += PS
    if PS <= 4096:
        assert seen == total            # each pixel exactly once per octet (lanes 0 .. PS-1 partition the pixels by px % PS)
    # border loop: frame pixels f = 0 .. B * nb - 1 of the [B][H+2][W+2] operand, walked with the same lane / stride
    Wq, Hq = W + 2, H + 2
    nb = 2 * Wq + 2 * H
    if B * nb <= 200000:
        frame = set()
        for f in range(B * nb):
            bb, r = divmod(f, nb)
            if r < Wq:
                yy, xx = 0, r
            elif r < 2 * Wq:
                yy, xx = Hq - 1, r - Wq
            else:
                k = r - 2 * Wq
                yy, xx = 1 + (k >> 1), (Wq - 1 if k & 1 else 0)
            assert (bb, yy, xx) not in frame
            frame.add((bb, yy, xx))
        want = {(bb, yy, xx) for bb in range(B) for yy in range(Hq) for xx in range(Wq) if yy in (0, Hq - 1) or xx in (0, Wq - 1)}
        assert frame == want
