"""CPU: the index mathematics of the 8-wave ping-pong 3x3 kernel (head_detector_amd/csrc/conv_pp.hip) restated in numpy and checked against what
the MFMA needs -- the LDS-DMA placement of a wave's private 10 x 10 halo, the swizzled fragment addresses of every tap, their ds_read_b128 bank behaviour, the
weight-ring / vmcnt schedule and the half-wave exchange of the register epilogue.  The kernel itself is tested on the GPU (test_conv_all_configs_vs_torch,
test_conv_persistent_multi_tile); this file pins the layout it is written against, so that a change of one constant fails here, without a GPU, with a name."""
import numpy as np

XU, PITCH = 7, 10  # 16-record LDS-DMA units of a halo stage; halo records per row (8 + 2)


def halo_stage(cb_chunk_of):
    """LDS image of one halo stage as (record, 16-byte slot) -> (hy, hx, channel chunk) for the 7 units a wave issues: lane l of unit u writes record
    u * 16 + (l >> 2), slot l & 3, and FETCHES chunk (l & 3) ^ (hy & 3) of halo pixel (hy, hx) (source-side swizzle)."""
    img = {}
    for u in range(XU):
        for lane in range(64):
            rec = u * 16 + (lane >> 2)
            hy, hx = divmod(rec, PITCH)
            slot = lane & 3
            img[(rec, slot)] = (hy, hx, cb_chunk_of(slot, hy)) if rec < 100 else None
    return img


def frag_byte(lane, j, ky, kx, h):
    """Byte offset (inside the halo stage) the kernel's B-fragment read uses: bofs[ky][h] + j * 2560 + kx * 64."""
    n32, hi = lane & 31, lane >> 5
    r = (n32 >> 3) + ky
    return (r * PITCH + (n32 & 7)) * 64 + (((2 * h + hi) ^ (r & 3)) * 16) + j * 2560 + kx * 64


def test_halo_fragment_reads_fetch_the_im2col_operand():
    img = halo_stage(lambda slot, hy: slot ^ (hy & 3))
    for j in range(2):
        for ky in range(3):
            for kx in range(3):
                for h in range(2):
                    for lane in range(64):
                        b = frag_byte(lane, j, ky, kx, h)
                        rec, slot = divmod(b, 64)
                        slot //= 16
                        hy, hx, chunk = img[(rec, slot)]
                        n32, hi = lane & 31, lane >> 5
                        row, col = 4 * j + (n32 >> 3), n32 & 7  # output pixel of this MFMA column inside the 8 x 8 sub-patch
                        assert (hy, hx) == (row + ky, col + kx)  # input pixel (row + ky - 1, col + kx - 1) of a pad-1 3x3 conv
                        assert chunk == 2 * h + hi  # k-slice [8 * (2h + hi), +8) of the 32-channel block: lanes 32-63 hold k 8..15 of the 32x32x16 MFMA


def test_fragment_reads_are_bank_conflict_free():
    # ds_read_b128 services a wave in four 16-lane groups (MI355X_MICROARCH.md, LDS table); within a group the 16 x 16 bytes must cover all 64 banks
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[g + 32 for g in grp] for grp in groups]
    for j in range(2):
        for ky in range(3):
            for kx in range(3):
                for h in range(2):
                    for grp in groups:
                        banks = {(frag_byte(lane, j, ky, kx, h) // 16) % 16 for lane in grp}
                        assert len(banks) == 16, (j, ky, kx, h)
    # weights: row = cout (lane & 31), slot = chunk ^ ((cout >> 2) & 3) (vgh_pack_conv_weights_host)
    for h in range(2):
        for grp in groups:
            banks = {(((lane & 31) * 64 + (((2 * h + (lane >> 5)) ^ (((lane & 31) >> 2) & 3)) * 16)) // 16) % 16 for lane in grp}
            assert len(banks) == 16


def test_weight_ring_and_vmcnt_schedule():
    """g tiles: in L(T) a wave issues [W(T+2), halo units, (residual touch)]; at the end of L(T) it waits until at most n_T of its LDS-DMA loads are outstanding.
    Simulate the in-order queue over three channel blocks and check (i) the weight unit of tap T+1 has landed at the end of L(T) -- one barrier before any wave
    reads it in L(T+1) --, (ii) the whole halo of the next block has landed at the end of L(8), (iii) a ring stage is never re-filled before the L phase that
    follows its last read (3 stages, prefetch distance 2)."""
    for xfront in (True, False):
        xn = lambda T: ((2 if T < 3 else 1 if T == 3 else 0) if xfront else (1 if T < 7 else 0))  # noqa: E731
        issued = []  # in-order queue of (kind, tag)
        for g in range(3 * 9):
            cb, T = divmod(g, 9)
            issued.append(("W", g + 2))
            issued += [("X", (cb + 1, u)) for u in range(xn(T))]
            n_T = xn((T + 8) % 9) + 1 + xn(T)  # the kernel's wait_vm<> immediate (without the optional residual touches)
            landed = issued[: len(issued) - n_T]
            if g >= 1:
                assert ("W", g + 1) in landed, (xfront, g)
            if T == 8:
                assert sum(1 for k, t in landed if k == "X" and t[0] == cb + 1) == XU
            # stage (g + 2) % 3 was last read in L(g - 1) (tap g - 1): refilled now, in L(g) -- strictly later
            assert (g + 2) % 3 == (g - 1) % 3
        assert sum(xn(T) for T in range(9)) == XU


def test_epilogue_half_wave_exchange_gives_eight_contiguous_couts():
    """Accumulator layout of v_mfma_f32_32x32x16: lane (n32, hi) holds, for register r = 4 q + e, cout 8 q + 4 hi + e of pixel n32.  The epilogue packs runs q = 2m
    (A) and 2m + 1 (B) to bf16 pairs and applies v_permlane32_swap(vdst = A, src = B): lanes 32-63 of A trade places with lanes 0-31 of B."""
    cout = lambda hi, q, e: 8 * q + 4 * hi + e  # noqa: E731
    for m in range(2):
        A = {hi: [cout(hi, 2 * m, e) for e in range(4)] for hi in (0, 1)}
        B = {hi: [cout(hi, 2 * m + 1, e) for e in range(4)] for hi in (0, 1)}
        A[1], B[0] = B[0], A[1]  # the swap
        for hi in (0, 1):
            got = A[hi] + B[hi]  # the 16-byte vector {pa0, pa1, pb0, pb1} this lane stores
            assert got == list(range(16 * m + 8 * hi, 16 * m + 8 * hi + 8))  # at channel offset 16 m + 8 hi: 32 contiguous bytes per pixel per instruction
        # the residual takes the inverse route: swap32(d0, d2), swap32(d1, d3) on the loaded vector d0..d3 = couts 16m + 8hi + (0,1),(2,3),(4,5),(6,7)
        d = {hi: [[16 * m + 8 * hi + 2 * i, 16 * m + 8 * hi + 2 * i + 1] for i in range(4)] for hi in (0, 1)}
        d[1][0], d[0][2] = d[0][2], d[1][0]
        d[1][1], d[0][3] = d[0][3], d[1][1]
        for hi in (0, 1):
            assert d[hi][0] + d[hi][1] == [cout(hi, 2 * m, e) for e in range(4)] and d[hi][2] + d[hi][3] == [cout(hi, 2 * m + 1, e) for e in range(4)]


def test_out_of_range_marker_survives_the_offsets_added_to_it():
    oob = 0xC0000000
    for add in (0, 224, 2 * 2048, (1 << 30) - 1):  # epilogue immediates, segment shift, the largest weight k-block offset
        assert 0x80000000 <= oob + add < (1 << 32)  # still beyond the 2 GiB descriptor range, not wrapped to a valid offset (0xFFFFFFF0 + 32 wraps to 16)
    assert (0xFFFFFFF0 + 32) % (1 << 32) == 16


def test_tile_bookkeeping_matches_the_launcher():
    # 8 x 8 sub-patches tile the maps of the 640 and 1280 configurations exactly; 8 per workgroup; the B = 64 quantisation quoted in DESIGN 3.9
    for W in (160, 80, 40, 320):
        assert W % 8 == 0
    nsp = 64 * (80 // 8) ** 2
    tiles = (nsp + 7) // 8
    assert tiles == 800 and np.isclose(tiles / 256, 3.125)


def test_just_in_time_halo_offsets_equal_the_full_pixel_address():
    """conv_pp.hip::unit_off builds a halo record's source offset from a wave-uniform base (byte offset of input pixel (y0 - 1, x0 - 1), "negative" on the top / left
    border) plus two 24-bit multiply-adds of the lane's (hy, hx), with hy = (hp * 205) >> 11 for hp / 10, in 32-bit wrap-around arithmetic.  Against the full address
    ((b H + iy) W + ix) pitch + coff of every in-range record, for sub-patches on every border, a view inside a concat buffer, and the largest row the launcher admits."""
    assert all((hp * 205) >> 11 == hp // 10 for hp in range(112))
    M32 = 1 << 32
    for H, W, pitch, coff in ((160, 160, 384, 96), (20, 20, 2048, 512), (8, 320, 768, 0), (40, 8191, 1024, 0)):
        assert 2 * W * pitch < (1 << 24)  # vgh_conv_pp_fits: one input row in 24 bits
        rowb, pixb = 2 * W * pitch, 2 * pitch
        nb = min(4, ((1 << 31) - 1) // (2 * H * W * pitch))  # the loaders' 2 GiB rule (vgh_conv_prepare)
        for b in (0, nb - 1):
            for y0 in (0, 8, (H - 1) // 8 * 8):
                for x0 in (0, 8, (W - 1) // 8 * 8):
                    base = 2 * (((b * H + y0 - 1) * W + x0 - 1) * pitch + coff)  # int; < 0 for b = 0, y0 = 0
                    for u in range(XU):
                        for lane in range(64):
                            hp = u * 16 + (lane >> 2)
                            hy = (hp * 205) >> 11
                            hx = hp - hy * 10
                            iy, ix = y0 - 1 + hy, x0 - 1 + hx
                            ok = hp < 100 and 0 <= iy < H and 0 <= ix < W
                            rel = hy * rowb + hx * pixb + (((lane & 3) ^ (hy & 3)) << 4)
                            assert hy < (1 << 24) and rowb < (1 << 24) and hx < (1 << 24) and pixb < (1 << 24)  # v_mul_u32_u24 operands
                            if ok:
                                full = 2 * (((b * H + iy) * W + ix) * pitch + coff) + (((lane & 3) ^ (hy & 3)) << 4)
                                assert (base % M32 + rel) % M32 == full and full < (1 << 31)


# ---- "s" tiles (r06, conv3x3_pp_kernel<..., GEO = 1>): the wave's 64 pixels are TWO 4-row x 8-column sub-patches with origins of their own ----
XU_S, JOFF_S = 8, 60 * 64  # units of a halo stage (two 6 x 10 halos = 120 of 128 records); bytes from pixel group 0's records to pixel group 1's


def halo_stage_s():
    """LDS image of a GEO = 1 halo stage: lane l of unit u writes record hp = 16 u + (l >> 2), slot l & 3, and fetches chunk (l & 3) ^ (hy & 3) of halo pixel
    (hy, hx) = divmod(hp % 60, 10) of sub-patch hp // 60 (conv_pp.hip::unit_off, GEO branch)."""
    img = {}
    for u in range(XU_S):
        for lane in range(64):
            hp = u * 16 + (lane >> 2)
            j, hl = divmod(hp, 60)
            hy, hx = divmod(hl, 10)
            img[(hp, lane & 3)] = (j, hy, hx, (lane & 3) ^ (hy & 3)) if hp < 120 else None
    return img


def frag_byte_s(lane, j, ky, kx, h):
    n32, hi = lane & 31, lane >> 5
    r = (n32 >> 3) + ky
    return (r * PITCH + (n32 & 7)) * 64 + (((2 * h + hi) ^ (r & 3)) * 16) + j * JOFF_S + kx * 64


def test_s_tiles_fragment_reads_fetch_the_im2col_operand_and_are_conflict_free():
    img = halo_stage_s()
    assert all((hl * 205) >> 11 == hl // 10 for hl in range(60))
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[g + 32 for g in grp] for grp in groups]
    for j in range(2):
        for ky in range(3):
            for kx in range(3):
                for h in range(2):
                    for lane in range(64):
                        rec, slot = divmod(frag_byte_s(lane, j, ky, kx, h), 64)
                        jj, hy, hx, chunk = img[(rec, slot // 16)]
                        n32, hi = lane & 31, lane >> 5
                        assert jj == j and (hy, hx) == ((n32 >> 3) + ky, (n32 & 7) + kx) and chunk == 2 * h + hi  # pixel (n32 >> 3, n32 & 7) of sub-patch j, tap (ky, kx)
                    for grp in groups:
                        assert len({(frag_byte_s(lane, j, ky, kx, h) // 16) % 16 for lane in grp}) == 16, (j, ky, kx, h)
    # eight units issued two per tap in taps 0 - 3; the counted waits of the g tiles with that schedule (test_weight_ring_and_vmcnt_schedule's simulation)
    xn = lambda T: 2 if T < 4 else 0  # noqa: E731
    issued = []
    for g in range(3 * 9):
        cb, T = divmod(g, 9)
        issued.append(("W", g + 2))
        issued += [("X", (cb + 1, u)) for u in range(xn(T))]
        landed = issued[: len(issued) - (xn((T + 8) % 9) + 1 + xn(T))]
        if g >= 1:
            assert ("W", g + 1) in landed
        if T == 8:
            assert sum(1 for k, t in landed if k == "X" and t[0] == cb + 1) == XU_S
    assert sum(xn(T) for T in range(9)) == XU_S
    # LDS budget of the 128-cout variant: 8 waves x 2 stages x 8 KB + 3 weight stages + dummy unit + a 4-KB bias vector (cout_pad <= 1024)
    assert 8 * 2 * XU_S * 1024 + 3 * 128 * 64 + 1024 + 4096 <= 160 * 1024


def test_s_tiles_cover_every_pixel_once_or_with_identical_duplicates():
    """4 x 8 sub-patches: the last one of a row / the last band is MOVED BACK inside the map (x0 = W - 8, y0 = H - 4), never cut: every output pixel is owned by at
    least one sub-patch, and a pixel owned twice is computed from the same 3 x 3 window both times (its value does not depend on the sub-patch origin)."""
    for H, W in ((20, 20), (10, 10), (40, 40), (21, 37), (4, 8), (7, 9)):
        nsx, nsy = (W + 7) // 8, (H + 3) // 4
        owners = np.zeros((H, W), dtype=int)
        for sy in range(nsy):
            for sx in range(nsx):
                y0, x0 = min(sy * 4, H - 4), min(sx * 8, W - 8)
                assert y0 >= 0 and x0 >= 0
                owners[y0:y0 + 4, x0:x0 + 8] += 1
        assert owners.min() >= 1
        assert owners.sum() == nsy * nsx * 32
    # the 20-wide maps of the 640 configuration: 15 sub-patches = 480 pixels for 400 (1.2 x); 8 x 8 patches cover 24 x 24 = 576 (1.44 x)
    assert ((20 + 7) // 8) * ((20 + 3) // 4) * 32 == 480 and ((20 + 7) // 8) ** 2 * 64 == 576
