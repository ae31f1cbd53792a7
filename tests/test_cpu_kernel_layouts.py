"""Index arithmetic of the round-3 kernels restated in numpy (no GPU): the LDS images and lane mappings the HIP sources rely on
are bijective, agree between the writing and the reading side, and keep the 16 lanes of a `ds_read_b128` group on 16 different
16-byte bank groups (gfx950: 64 banks x 4 B = 16 groups of 16 B).  The formulas are copied from the kernels' comments; the
GPU tests check the kernels themselves."""
import numpy as np


def _bank_groups_distinct(slots16):
    """slots16: 16-byte slot indices read by the 16 lanes of one ds_read_b128 group."""
    return len({int(s) % 16 for s in slots16}) == len(slots16)


# lane groups one LDS cycle serves (MI355X_MICROARCH.md, section LDS): ds_read_b128 in 4 groups of 16 lanes, ds_write_b128 in 8 of 8
_READ_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
_READ_B128_GROUPS += [[ln + 32 for ln in g] for g in _READ_B128_GROUPS]


def test_winograd_epilogue_read_side_mapping():
    """csrc/conv3_wino.hip epilogue (round 6).  Exchange region = [f 4][col 128][8 items of 16 B], item k of column col at slot
    k ^ sigma(col), sigma = ((col & 1) << 2) | ((col >> 1) & 3), frequency f at f * 1024 + (0, 0, 8, 16)[f] items.
    Write side = MFMA layout (lane (j, h): column ct * 32 + j, rows 8 q + 4 h + e = item 2 q + h of wave f's region).
    Read side = lane (half, x, cgp) x slot yr: output position (row yr, x), channels 4 (2 cgp + half) ..+3; an output reads THREE
    frequencies: y0 (x even): m0, m1, m2; y1 (x odd): m2, m1, m3."""
    OF = (0, 0, 8, 16)

    def sigma(col):
        return ((col & 1) << 2) | ((col >> 1) & 3)

    def item(f, col, k):
        return f * 1024 + OF[f] + col * 8 + (k ^ sigma(col))

    # the map (f, col, k) -> item is injective and the frequencies do not overlap
    all_items = {item(f, col, k) for f in range(4) for col in range(128) for k in range(8)}
    assert len(all_items) == 4 * 128 * 8 and max(all_items) < 4 * 1024 + 16
    lanes = np.arange(64)
    j, h = lanes & 31, lanes >> 5
    # write side as the kernel computes it: base = f * 1024 + OF[f] + j * 8, + ((2 q + h) ^ sj), + ct * 256
    for f in range(4):
        for q in range(4):
            for ct in range(4):
                got = f * 1024 + OF[f] + j * 8 + ((2 * q + h) ^ (((j & 1) << 2) | ((j >> 1) & 3))) + ct * 256
                want = np.array([item(f, ct * 32 + int(jj), 2 * q + int(hh)) for jj, hh in zip(j, h)])
                assert (got == want).all()
                for g0 in range(0, 64, 8):           # ds_write_b128: 8 consecutive lanes per LDS cycle, 8 slots of 16 B (32 banks)
                    assert len({int(v) % 8 for v in got[g0:g0 + 8]}) == 8
    # read side as the kernel computes it
    half, xq, cgp = lanes & 1, (lanes >> 1) & 7, lanes >> 4
    cq, prr, odd = 2 * cgp + half, xq >> 1, xq & 1
    rx0 = cq ^ (((prr & 1) << 2) | (prr >> 1))
    offA = np.where(odd == 1, 2 * 1024 + 8, 0)
    offC = np.where(odd == 1, 3 * 1024 + 16, 2 * 1024 + 8)
    covered = set()
    for wid in range(4):
        for yr in range(8):
            base = (wid * 32 + prr) * 8 + yr * 32 + (rx0 ^ 2 if yr & 1 else rx0)
            col = wid * 32 + yr * 4 + prr
            for off, fsel in ((offA, np.where(odd == 1, 2, 0)), (np.full(64, 1024), np.full(64, 1)), (offC, np.where(odd == 1, 3, 2))):
                got = base + off
                want = np.array([item(int(f), int(c), int(k)) for f, c, k in zip(fsel, col, cq)])
                assert (got == want).all()
                for g in _READ_B128_GROUPS:          # 16 lanes per LDS cycle on 16 slots of 16 B: distinct slots, or the same item (broadcast)
                    slots = {}
                    for v in got[g]:
                        slots.setdefault(int(v) % 16, set()).add(int(v))
                    assert all(len(v) == 1 for v in slots.values())
            covered |= {(wid, yr, int(x), int(c)) for x, c in zip(xq, cq)}
    assert len(covered) == 4 * 8 * 8 * 8             # every (plane, row, x, channel quad) of the round's 32 rows x 256 positions once
    # a store instruction (one slot yr): per channel-group pair the 16 lanes (half, x) write 16 consecutive 16-byte pieces = one
    # whole 256-byte tile row of the F32B tensor ([C/8][P][8] floats: position stride 32 B, the quad's half 16 B)
    byte_off = xq * 32 + half * 16
    for g0 in range(0, 64, 16):
        assert sorted(byte_off[g0:g0 + 16].tolist()) == list(range(0, 256, 16)) and len(set(cgp[g0:g0 + 16].tolist())) == 1
    # statistics: row_ror 8, 4, 2 inside a DPP row of 16 sum the 8 lanes that share `half` (same channel quad, the 8 x of a row)
    for ln in range(64):
        row = ln & ~15
        mates = {row | ((ln + r) & 15) for r in range(0, 16, 2)}
        assert {int(cq[m]) for m in mates} == {int(cq[ln])} and len({int(xq[m]) for m in mates}) == 8


def test_stride2_slab_image_and_fragment_reads():
    """csrc/conv3_s2.hip: slab row image = 9 even-x slots, 8 odd-x slots, 3 pad (20 slots); a fragment read of tap kw by the
    output at x (8 per tile row) must find the input position 2 x + kw, and 16 lanes (8 x by 2 output rows) must not conflict."""
    YH, ROW = 17, 20

    def commit_slot(zl, hy, hx):
        return (zl * YH + hy) * ROW + (hx & 1) * 9 + (hx >> 1)

    def read_slot(wc, cm, j, kh, kw):
        return (wc * YH + 2 * (cm * 4 + (j >> 3)) + kh) * ROW + (kw & 1) * 9 + (kw >> 1) + (j & 7)

    slots = {commit_slot(zl, hy, hx) for zl in range(4) for hy in range(17) for hx in range(17)}
    assert len(slots) == 4 * 17 * 17 and max(slots) < 4 * YH * ROW
    pads = {(zl * YH + hy) * ROW + p for zl in range(4) for hy in range(17) for p in (17, 18, 19)}
    assert not (slots & pads)                         # items beyond the slab are parked in pad slots
    for wc in range(4):
        for cm in range(2):
            for kh in range(3):
                for kw in range(3):
                    for j in range(32):
                        y, x = cm * 4 + (j >> 3), j & 7
                        assert read_slot(wc, cm, j, kh, kw) == commit_slot(wc, 2 * y + kh, 2 * x + kw)
                    for g in range(2):
                        assert _bank_groups_distinct([read_slot(wc, cm, j, kh, kw) for j in range(16 * g, 16 * g + 16)])


def test_head_and_stem_halo_images():
    """csrc/conv3_head.hip / conv3_stem.hip: halo position p = hz * 80 + hy * 8 + hx of a 6 x 10 x 8 halo (no x halo: the taps
    along x are folded into rows / channels); the fragment of tap (kd, kh) for output (z, y, x) reads halo (z + kd, y + kh, x)."""
    for j in range(32):
        for wc in range(4):
            for half in range(2):                     # head: y half per wave; stem: column tile cm
                for kd in range(3):
                    for kh in range(3):
                        y, x = half * 4 + (j >> 3), j & 7
                        slot = ((wc * 10 + half * 4 + (j >> 3)) * 8 + (j & 7)) + (kd * 80 + kh * 8)
                        assert slot == (wc + kd) * 80 + (y + kh) * 8 + x
    for kd in range(3):
        for kh in range(3):
            for g in range(2):
                assert _bank_groups_distinct([(j >> 3) * 8 + (j & 7) + kd * 80 + kh * 8 for j in range(16 * g, 16 * g + 16)])


def test_tiled_weight_packer_covers_every_item_once():
    """csrc/pack_batch.hip md_pack_tiled_kernel: blocks (16-row block rb, chunk cc) x tasks (tap, channel group g, row rr) write
    items base and base + nt of the WPK layout [rt][cc][tap][kg][part 2][nt][8]: every item exactly once."""
    rows, kdim, taps, nt, kc = 136, 96, 27, 128, 32
    kg, ncc = kc // 8, -(-kdim // kc)
    nrb = -(-rows // nt) * (nt // 16)
    n_items = -(-rows // nt) * ncc * taps * kg * 2 * nt
    hit = np.zeros(n_items, dtype=np.int32)
    for lb in range(nrb * ncc):
        rb, cc = lb % nrb, lb // nrb
        for task in range(taps * kg * 16):
            rr, tg = task % 16, task // 16
            g, tap = tg % kg, tg // kg
            row = rb * 16 + rr
            rt, rin = row // nt, row % nt
            base = ((((rt * ncc + cc) * taps + tap) * kg + g) * 2) * nt + rin
            hit[base] += 1
            hit[base + nt] += 1
    assert hit.min() == 1 and hit.max() == 1


def test_attention_v_image_matches_the_p_key_order():
    """csrc/attention.hip: thread c stores, for key step ks, plane p and lane half h, the 16-byte item [ks][p][h][c] =
    {group 2ks half h, group 2ks + 1 half h} of the tile's V items [kg][p][c] (8 keys each: half 0 = keys 0..3, half 1 =
    keys 4..7); lane (j, h) of row tile rt reads item [ks][p][h][32 rt + j] and must find the keys of MFMA k-slots
    (h, i) = 16 ks + 8 (i / 4) + 4 h + i % 4 (the order the S^T accumulator rows hand P over in).  One ds_read_b128 lane
    group = 16 consecutive 16-byte items."""
    C = 256

    def item(ks, p, h, c):
        return ((ks * 2 + p) * 2 + h) * C + c

    written = {}
    for c in range(C):
        for ks in range(2):
            for p in range(2):
                for h in range(2):
                    keys = [8 * (2 * ks) + 4 * h + e for e in range(4)] + [8 * (2 * ks + 1) + 4 * h + e for e in range(4)]
                    assert item(ks, p, h, c) not in written
                    written[item(ks, p, h, c)] = (p, c, keys)
    assert sorted(written) == list(range(2 * 2 * 2 * C))            # the 32 KB image is covered exactly once
    lanes = np.arange(64)
    j, h = lanes & 31, lanes >> 5
    for rt in range(C // 32):
        for ks in range(2):
            for p in range(2):
                slots = [item(ks, p, int(hh), rt * 32 + int(jj)) for jj, hh in zip(j, h)]
                for ln, s in enumerate(slots):
                    pp, cc, keys = written[s]
                    assert pp == p and cc == rt * 32 + (ln & 31)
                    assert keys == [16 * ks + 8 * (i // 4) + 4 * (ln >> 5) + i % 4 for i in range(8)]
                for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                            [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
                    for half in (0, 32):
                        assert _bank_groups_distinct([slots[half + g] for g in grp])


def test_winograd_f8_halo_request_schedule_three_groups_in_flight():
    """csrc/conv3_wino.hip, pair-step loop with W8_HG = 3: a body = 9 pair-steps over two chunks; pair-step u stores the group of four
    halo pieces st_base(u) .. + 3 of chunk c0 + 1 (u <= 3, buffer 1) or c0 + 2 (u >= 5, buffer 0) out of slot slot3(u), and requests
    the group base_of(u) .. + 3 of chunk c0 + 1 / + 2 / + 3 into slot slot3(u).  Replayed over several bodies: every stored group was
    requested exactly three pair-steps earlier, for the chunk and the pieces the store expects, into the slot it is read from, no slot
    is refilled while it still holds an unstored group, at most three groups are in flight, and every chunk's 15 pieces reach LDS."""
    slot3 = lambda u: 0 if u in (0, 3, 6) else (1 if u in (2, 5, 8) else 2)          # noqa: E731
    base_of = lambda u: 12 if u == 0 else ((u - 2) * 4 if 2 <= u <= 5 else ((u - 6) * 4 if u >= 6 else None))   # noqa: E731
    req_chunk = lambda u, c0: c0 + 1 if u == 0 else (c0 + 2 if u <= 5 else c0 + 3)  # noqa: E731
    st_base = lambda u: u * 4 if u <= 3 else ((u - 5) * 4 if u >= 5 else None)       # noqa: E731
    st_chunk = lambda u, c0: c0 + 1 if u <= 3 else c0 + 2                            # noqa: E731
    nbodies = 5
    # prologue: chunk 0 whole into buffer 0; chunk 1's pieces 0..11 requested into the slots of the stores at pair-steps 0, 1, 2
    slots = {slot3(g): dict(chunk=1, pieces=list(range(4 * g, 4 * g + 4)), t=-3 + g) for g in range(3)}
    stored = {0: set(range(15))}
    t = 0
    for body in range(nbodies):
        c0 = 2 * body
        for u in range(9):
            sb = st_base(u)
            if sb is not None:                       # pass 0 of groups 0..2: the stores
                ent = slots.pop(slot3(u))
                want = [p for p in range(sb, sb + 4) if p < 15]
                assert ent["chunk"] == st_chunk(u, c0) and [p for p in ent["pieces"] if p < 15] == want, (body, u, ent)
                assert t - ent["t"] == 3, (body, u)
                stored.setdefault(ent["chunk"], set()).update(want)
            lb = base_of(u)
            if lb is not None:                       # pass 1 of group 2: the requests
                assert slot3(u) not in slots, (body, u)          # the slot was emptied by this pair-step's stores (or earlier)
                slots[slot3(u)] = dict(chunk=req_chunk(u, c0), pieces=list(range(lb, lb + 4)), t=t)
            assert len(slots) <= 3
            t += 1
        # the chunks this body's MFMAs read next are complete: c0 + 1 from pair-step 4 on, c0 + 2 at the start of the next body
        assert stored[c0 + 1] == set(range(15)) and stored[c0 + 2] == set(range(15))


B128_READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_READ_GROUPS += [[l + 32 for l in g] for g in B128_READ_GROUPS]      # MI355X_MICROARCH.md, LDS table: ds_read_b128 lane groups


def test_operand_pass_parity_split_image_is_conflict_free():
    """csrc/wino_prep2.hip (round 5): the 256 activated positions of a channel group live in LDS as [parity][128][12 floats] with the
    odd region 16 floats further (p2_slot).  Phase 1: thread = position, two ds_write_b128 (contiguous 8-lane groups, 32 banks);
    phase 2: thread = pair pi reads positions 2 pi - 1 + k, two ds_read_b128 each (64 banks, the guide's 16-lane groups and --
    for good measure -- contiguous ones).  The image is a bijection and no group has two lanes on one 16-byte slot; the round-4
    image (positions in one run, 12 floats apart) is 2-way conflicted on every read: asserted too, so the test means something."""
    STRIDE, REGION = 12, 128 * 12 + 16

    def slot(p):
        return (p & 1) * REGION + (p >> 1) * STRIDE

    words = {slot(p) + e for p in range(256) for e in range(8)}
    assert len(words) == 256 * 8 and max(words) < 2 * REGION and all(slot(p) % 4 == 0 for p in range(256))
    # phase 1 stores
    for half in (0, 4):
        for g0 in range(0, 256, 8):
            banks = [((slot(t) + half) // 4) % 8 for t in range(g0, g0 + 8)]          # 16-byte slot of the 128-byte (32-bank) row
            assert len(set(banks)) == 8, (g0, banks)
    # phase 2 reads: wave w of the workgroup holds pairs 64 (w & 1) .. + 63
    groups = B128_READ_GROUPS + [list(range(16 * g, 16 * g + 16)) for g in range(4)]
    for k in range(4):
        for half in (0, 4):
            for wave in range(2):
                for grp in groups:
                    ps = [2 * (64 * wave + ln) - 1 + k for ln in grp]
                    ps = [p for p in ps if 0 <= p < 256]
                    new = [((slot(p) + half) // 4) % 16 for p in ps]
                    assert len(set(new)) == len(new), (k, wave, grp)
                    old = [((p * STRIDE + half) // 4) % 16 for p in ps]
                    assert len(set(old)) <= (len(old) + 1) // 2 + 1       # the replaced image: period 8 -> 2-way
