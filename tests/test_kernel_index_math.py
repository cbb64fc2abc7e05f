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


# ---- epilogue of the 256-wide kernels (csrc/igemm_sw_common.h, round 4) -----------------------------------------------------------
SW_EPI_PITCH = 1088         # LDS bytes per 4-row block of the residual landing zone


def acc_row(lk, r):
    """row (inside a 32-row MFMA tile) of accumulator register r of a lane in half-wave lk (v_mfma_f32_32x32x16_f16 C layout)"""
    return 4 * lk + (r & 3) + 8 * (r >> 2)


@pytest.mark.parametrize("NQ", [1, 2], ids=["dw: 64 x 128 wave tile", "sw: 128 x 128 wave tile"])
def test_residual_landing_zone_addressing(NQ):
    """The fp16 residual of a wave tile (64 NQ rows x 128 columns) lands in LDS by 16 NQ LDS-DMA instructions - block b = rows 4b .. 4b+3,
    lane l fetches 16 bytes (row l >> 4 of the block, columns 8 (l & 15) ..) and lands at b * SW_EPI_PITCH + 16 l - and every lane then
    reads, for MFMA tile (i, j) and register pair k, its own column's rows (2k, 2k+1) at CONSTANT offsets from one base register.
    Restated element by element: every read picks up exactly the element the accumulator register holds the output of; the two
    half-waves of a read instruction touch disjoint LDS banks (the 64-byte skew of the pitch)."""
    rows, cols = 64 * NQ, 128
    lds = {}                                    # LDS byte address (even) -> (row, col) of the fp16 element stored there
    for b in range(16 * NQ):
        for lane in range(64):
            row, c0 = 4 * b + (lane >> 4), (lane & 15) * 8
            for h in range(8):
                addr = b * SW_EPI_PITCH + lane * 16 + 2 * h
                assert addr not in lds
                lds[addr] = (row, c0 + h)
    assert len(lds) == rows * cols and max(lds) < 16 * NQ * SW_EPI_PITCH
    for i in range(2 * NQ):
        for j in range(4):
            for k in range(8):
                for e in range(2):
                    off = (8 * i + 2 * (k >> 1)) * SW_EPI_PITCH + 2 * (k & 1) * 256 + j * 64 + e * 256      # the kernel's `o` and `o + 128` halves
                    banks = {0: set(), 1: set()}
                    for lk in range(2):
                        for lr in range(32):
                            addr = lk * SW_EPI_PITCH + lr * 2 + off
                            want = (32 * i + acc_row(lk, 2 * k + e), 32 * j + lr)
                            assert lds[addr] == want, (i, j, k, e, lk, lr)
                            banks[lk].add((addr // 4) % 64)
                    assert len(banks[0]) == 16 and len(banks[1]) == 16 and not (banks[0] & banks[1])
                    assert not ({x % 32 for x in banks[0]} & {x % 32 for x in banks[1]})           # also disjoint on a 32-bank LDS


def test_fp16_pair_store_selectors():
    """OUT16: a lane converts its pair (rows 2k, 2k+1 of ITS column) with one v_cvt_pk_f16_f32 (low half = row 2k), takes the packed pair
    of the lane of the neighbouring column through DPP quad_perm [1,0,3,2] and picks two halves with ONE v_perm_b32 whose selector
    depends on the lane's parity: the even lane stores row 2k, columns (lr, lr+1), the odd lane row 2k+1, columns (lr-1, lr) - at the byte
    offset ((4 lk + odd) * ld + lr - odd) * 2 from the pair's row base.  v_perm_b32 D = bytes of {S0 (partner), S1 (own)}: selector byte
    values 0-3 pick S1's bytes, 4-7 pick S0's."""
    def perm(s0, s1, sel):
        src = [(s1 >> (8 * n)) & 0xFF for n in range(4)] + [(s0 >> (8 * n)) & 0xFF for n in range(4)]
        return sum(src[(sel >> (8 * n)) & 0xFF] << (8 * n) for n in range(4))

    SEL = {0: 0x05040100, 1: 0x03020706}
    ld = 256
    tile = {}                                    # (row, col) -> 16-bit pattern standing for the fp16 value
    val = lambda row, col: (row * 251 + col * 7 + 1) & 0xFFFF
    packed = lambda lk, lr, k: val(acc_row(lk, 2 * k), lr) | (val(acc_row(lk, 2 * k + 1), lr) << 16)
    stored = {}
    for k in range(8):
        rowb = 2 * (k & 1) + 8 * (k >> 1)        # the pair's row inside the tile for lk = 0 (wave-uniform part of the address)
        for lk in range(2):
            for lr in range(32):
                odd = lr & 1
                own, partner = packed(lk, lr, k), packed(lk, lr ^ 1, k)
                d = perm(partner, own, SEL[odd])
                byte = (rowb * ld) * 2 + ((4 * lk + odd) * ld + lr - odd) * 2
                row, col = byte // 2 // ld, byte // 2 % ld
                for h in range(2):
                    assert (row, col + h) not in stored
                    stored[(row, col + h)] = (d >> (16 * h)) & 0xFFFF
    assert len(stored) == 32 * 32
    assert all(v == val(r, c) for (r, c), v in stored.items())


def test_column_record_order_is_two_chains_per_lane():
    """The GroupNorm column records of every tile kernel sum a lane's 16 values of a 32-row MFMA tile as TWO chains - even and odd
    accumulator registers, i.e. the first and second row of every register pair - then add the chains, then the partner half-wave, then
    the two MFMA tiles of a 64-row record.  The packed fp32 epilogue (two rows per instruction) and the scalar epilogues of the other
    kernels must spell the same order: restated in float32 and compared bit for bit against a per-chain evaluation."""
    rng = np.random.default_rng(0)
    v = rng.standard_normal((64, 1)).astype(np.float32) * 3          # one column of a 64-row record
    def scalar_form(vals):                       # igemm.hip / igemm_h2.hip / igemm_pp_common.h: cs2[r & 1] += v
        tot = np.float32(0)
        parts = []
        for i in range(2):                       # the two 32-row MFMA tiles
            half = []
            for lk in range(2):
                cs2 = [np.float32(0), np.float32(0)]
                for r in range(16):
                    cs2[r & 1] = np.float32(cs2[r & 1] + vals[32 * i + acc_row(lk, r), 0])
                half.append(np.float32(cs2[0] + cs2[1]))
            parts.append(np.float32(half[0] + half[1]))
        return np.float32(parts[0] + parts[1])
    def packed_form(vals):                       # igemm_sw_common.h: s2 += {v[2k], v[2k+1]} (v_pk_add_f32), then s2.x + s2.y
        parts = []
        for i in range(2):
            half = []
            for lk in range(2):
                s2 = np.zeros(2, dtype=np.float32)
                for k in range(8):
                    pair = np.array([vals[32 * i + acc_row(lk, 2 * k), 0], vals[32 * i + acc_row(lk, 2 * k + 1), 0]], dtype=np.float32)
                    s2 = (s2 + pair).astype(np.float32)
                half.append(np.float32(s2[0] + s2[1]))
            parts.append(np.float32(half[0] + half[1]))
        return np.float32(parts[0] + parts[1])
    assert scalar_form(v).tobytes() == packed_form(v).tobytes()
    assert scalar_form(v * v).tobytes() == packed_form(v * v).tobytes()
    assert abs(float(scalar_form(v)) - float(v.astype(np.float64).sum())) < 1e-4


# ---- one-pass fp16 attention (csrc/attention.hip, H1 form, round 4) --------------------------------------------------------------
def swz64(row, slot):
    return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4)


@pytest.mark.parametrize("cfg", [(64, 4), (64, 2), (256, 4)], ids=["D=64 QW=4", "D=64 QW=2", "D=256 QW=4"])
def test_one_pass_attention_staging_and_fragments(cfg):
    """attn_flash_kernel<QW, D, H1 = true>: a stage is SL * 32 rows of K (slice-major: row = sl * 32 + key, 64 bytes = 32 d-values of that key,
    read IN PLACE from the fp16 qkv tensor) followed by D rows of V^T (32 key positions each, written by attn_pack_vt).  Restated: which
    16-byte unit every LDS-DMA lane fetches and where it lands; which unit every MFMA fragment read of S^T = K Q^T and O^T += V^T P^T picks
    up; that the d-range of a K fragment equals the d-range of the Q fragment it is multiplied with; and that the key ORDER inside a V^T row
    (attn_pack_vt) is the order in which the S^T accumulator registers hand the probabilities back as the B operand."""
    D, QW = cfg
    NT, SL, KB = QW * 64, D // 32, 32
    KT, VT = SL * KB * 64, D * 64
    STAGE = KT + VT
    PIECES = STAGE // (NT * 16)
    RPP = NT // 4
    assert (SL * KB) % RPP == 0 and PIECES * NT * 16 == STAGE
    lds = {}                                     # byte offset of a 16-byte unit in the stage -> ("K", key, d0) | ("V", dd, pos0)
    for tid in range(NT):
        wave, lane = tid // 64, tid % 64
        r_in_piece, ps = tid // 4, tid % 4
        for i in range(PIECES):
            row = i * RPP + r_in_piece
            ls = ps ^ ((row >> 2) & 3)
            dst = wave * 1024 + i * RPP * 64 + lane * 16             # M0 = stage + wave's first row of the piece; lane l lands at + 16 l
            assert dst == row * 64 + ps * 16                         # i.e. physical slot ps of LDS row `row`
            if row < SL * KB:
                sl, key = divmod(row, KB)
                lds[dst] = ("K", key, sl * 32 + ls * 8)              # source: qkv row of the key, k channels sl * 32 + 8 ls ..
            else:
                lds[dst] = ("V", row - SL * KB, ls * 8)              # source: V^T row dd, key positions 8 ls ..
    assert len(lds) == STAGE // 16
    for lk in range(2):
        for lr in range(32):
            for ks in range(D // 16):                                # S^T = K_blk Q^T: A = K rows (key lr), B = Q (query lr), k = 8 d-values
                sl, st = ks >> 1, ks & 1
                unit = lds[sl * (KB * 64) + swz64(lr, st * 2 + lk)]
                q_d0 = (ks * 2 + lk) * 8                             # the Q fragment: bytes (ks * 2 + lk) * 16 of the query's row
                assert unit == ("K", lr, q_d0), (lk, lr, ks, unit)
            for g in range(2):                                       # O^T += V^T_blk P^T: A = V^T rows (d), B = P registers 8 g .. 8 g + 7
                for t in range(D // 32):
                    unit = lds[KT + swz64(t * 32 + lr, g * 2 + lk)]
                    assert unit == ("V", t * 32 + lr, (g * 2 + lk) * 8)
    # key order of a V^T row: position pos holds key kk(pos) (attn_pack_vt); the B operand of key group g supplied by half-wave lk is the
    # accumulator registers r = 8 g + j, whose keys are acc_row(lk, r) - position (2 g + lk) * 8 + j of the row must hold exactly that key
    def kk(pos):
        p16 = pos & 15
        return (pos & 16) + ((p16 >> 3) & 1) * 4 + ((p16 >> 2) & 1) * 8 + (p16 & 3)
    assert sorted(kk(p) for p in range(32)) == list(range(32))
    for g in range(2):
        for lk in range(2):
            for j in range(8):
                assert kk((2 * g + lk) * 8 + j) == acc_row(lk, 8 * g + j)
