"""CPU emulation of the INDEX SCHEME of csrc/ssdhip_convh.hip (development aid, no GPU needed).

The slab kernel's correctness rests on integer bookkeeping that is easy to get wrong and expensive to debug on a GPU box:
the padded position grid (one dummy column per row, one dummy row per image), the slab <-> position map, the source-side
chunk permutation of the LDS-DMA image, the per-tap row displacement with its recomputed swizzle, the weight ring, the MFMA
fragment <-> (channel, position) map and the epilogue's position -> pixel map.  This script replays exactly that bookkeeping
lane by lane in NumPy -- LDS as a byte-addressed array of 16-byte chunks, an LDS-DMA piece as "lane L writes chunk at dst + 16 L",
an MFMA as the documented 32x32x16 operand / result layout -- and compares the result with a direct convolution.

    python tools/emulate_convh.py            # a few small shapes, ~10 s
"""
import itertools
import sys

import numpy as np

BN, BM, WST = 256, 128, 128 * 128


def emulate(B, H, W, Cin, Cout, NW, SPW, seed=0, G=None, csh=0, pool=False):
    """G: None = one workgroup per tile; an int (multiple of 8) = that many PERSISTENT workgroups, each walking over its tiles with
    one LDS image and requesting the next tile's first slab / weights during the last slice of the current one (MODE bit 128)."""
    rng = np.random.RandomState(seed)
    x = rng.randint(-3, 4, size=(B, H, W, Cin)).astype(np.float32)
    wt = rng.randint(-2, 3, size=(Cout, 3, 3, Cin)).astype(np.float32)
    xf = x.reshape(-1)                               # element units; the kernel's byte offsets are 2x these
    wf = wt.reshape(-1)
    D = NW
    SLAB0, SLB = NW * WST, SPW * 8192
    lds_bytes = SLAB0 + 2 * SLB
    W1, H1 = W + 1, H + 1
    Q = B * H1 * W1
    q_tiles = (Q + BN - 1) // BN
    G2 = csh != 0
    TC, TR = (1 << csh, 256 >> csh) if G2 else (1, 1)
    SC2 = TC + 2
    if G2:                                               # 2-D tiles of TR x TC pixels of one image
        WT, HT = (W + TC - 1) // TC, (H + TR - 1) // TR
        q_tiles = B * HT * WT
        assert (TR + 2) * SC2 <= 64 * SPW
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    n_tiles = Cout // BM
    csteps = Cin // 64
    assert Cin % 128 == 0 and Cout % 128 == 0 and (G2 or 256 + 2 * W + 4 <= 64 * SPW) and SPW + D <= 10
    y = np.full((B, Ho, Wo, Cout) if pool else (B, H, W, Cout), np.nan, np.float32)
    OOB = None

    total_ids = ((q_tiles + 7) // 8) * n_tiles * 8

    def tile_of(i):
        xcd, slot = i & 7, i >> 3
        qt = (slot // n_tiles) * 8 + xcd
        return ((qt if G2 else qt * BN), (slot % n_tiles) * BM) if (i < total_ids and qt < q_tiles) else None

    def tile_origin(qt):
        wt, r = qt % WT, qt // WT
        b = r // HT
        return b, (r - b * HT) * TR, wt * TC

    SP = (TR + 2) * SC2 if G2 else BN + 2 * W + 4
    stride = G if G else total_ids
    for wg in range(min(stride, total_ids)):
        if tile_of(wg) is None:
            continue
        lds = np.full((lds_bytes // 16, 8), np.nan, np.float32)      # chunks of 8 elements (16 bytes); chunk address = byte address / 16

        def dma(src, voff_elems, soff_elems, dst_byte):               # one wave-wide piece: per-lane element offsets or OOB
            for lane in range(64):
                v = voff_elems[lane]
                c = dst_byte // 16 + lane
                if v is OOB:
                    lds[c] = 0.0
                else:
                    o = v + soff_elems
                    lds[c] = src[o:o + 8]

        def make_xoff(q0):
            xo = np.empty((8, 64, SPW), object)
            for wave in range(8):
                for lane in range(64):
                    row0 = wave * 8 + (lane >> 3)
                    if G2:
                        b, h0, w0 = tile_origin(q0)
                        for k in range(SPW):
                            row = row0 + 64 * k
                            j = (lane & 7) ^ ((row >> 1) & 7)
                            sr, sc = row // SC2, row % SC2
                            hh, ww = h0 - 1 + sr, w0 - 1 + sc
                            ok = row < SP and 0 <= hh < H and 0 <= ww < W
                            xo[wave, lane, k] = (((b * H + hh) * W + ww) * Cin + j * 8) if ok else OOB
                        continue
                    q = q0 - (W + 2) + row0
                    b = h = w = 0
                    if q >= 0:
                        b = q // (H1 * W1); r = q - b * H1 * W1; h = r // W1; w = r - h * W1
                    else:
                        w = q
                    for k in range(SPW):
                        row = row0 + 64 * k
                        j = (lane & 7) ^ ((row >> 1) & 7)
                        ok = 0 <= w < W and h < H and q < Q and row < SP
                        xo[wave, lane, k] = (((b * H + h) * W + w) * Cin + j * 8) if ok else OOB
                        q += 64; w += 64
                        while w >= W1:
                            w -= W1; h += 1
                            if h == H1:
                                h = 0; b += 1
            return xo

        woff = np.empty((8, 64, 2), object)                           # tile invariant: the tile's first channel rides in the scalar offset
        for wave in range(8):
            for lane in range(64):
                for i in range(2):
                    row = (i * 8 + wave) * 8 + (lane >> 3)
                    j = (lane & 7) ^ ((row >> 1) & 7)
                    woff[wave, lane, i] = row * 9 * Cin + j * 8

        def issue_w(co, cs, tap, stage):
            for wave in range(8):
                for i in range(2):
                    dma(wf, woff[wave, :, i], (co * 9 + tap) * Cin + cs * 64, stage * WST + wave * 1024 + i * 8192)

        def issue_slab_piece(xo, cs_parity, soff, k):
            for wave in range(8):
                dma(xf, xo[wave, :, k], soff, SLAB0 + cs_parity * SLB + wave * 1024 + k * 8192)

        def read_frags(cs, tap, stage):
            """-> fa[wave][lane][kk][ci] (8 elems), fb[wave][lane][kk][pi]"""
            fa = np.empty((8, 64, 4, 2, 8), np.float32)
            fb = np.empty((8, 64, 4, 2, 8), np.float32)
            toff = (tap // 3) * (SC2 if G2 else W1) + tap % 3
            sl = SLAB0 + (cs & 1) * SLB
            for wave in range(8):
                wm, wn = wave >> 2, wave & 3
                for lane in range(64):
                    r31, khalf = lane & 31, lane >> 5
                    rowa = wm * 64 + r31
                    slot2 = wn * 32 + r31
                    prow = (2 * (slot2 >> csh)) * SC2 + (slot2 & (TC - 1)) if G2 else wn * 64 + r31
                    pistep = SC2 if G2 else 32
                    for kk in range(4):
                        abase = rowa * 128 + (((2 * kk + khalf) ^ ((rowa >> 1) & 7)) << 4)
                        for ci in range(2):
                            fa[wave, lane, kk, ci] = lds[(stage * WST + abase + ci * 4096) // 16]
                        for pi in range(2):
                            row = prow + pi * pistep + toff
                            rb = sl + (row << 7)
                            re = (((row >> 1) & 7) ^ khalf) << 4
                            fb[wave, lane, kk, pi] = lds[(rb + (re ^ (kk << 5))) // 16]
            return fa, fb

        def mfma_step(acc, fa, fb):
            # v_mfma_f32_32x32x16: A[row = r31][k = 8 khalf ..], B[k = 8 khalf ..][col = r31]; D[row = 8 g + 4 khalf + e][col = r31] in v = 4 g + e
            assert not np.isnan(fa).any() and not np.isnan(fb).any(), "a fragment read hit LDS nobody wrote (or the epilogue's stage)"
            for wave in range(8):
                for kk in range(4):
                    for ci in range(2):
                        A = np.zeros((32, 16), np.float32)
                        for lane in range(64):
                            A[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = fa[wave, lane, kk, ci]
                        for pi in range(2):
                            Bm = np.zeros((16, 32), np.float32)
                            for lane in range(64):
                                Bm[8 * (lane >> 5):8 * (lane >> 5) + 8, lane & 31] = fb[wave, lane, kk, pi]
                            Dm = A @ Bm
                            for lane in range(64):
                                r31, khalf = lane & 31, lane >> 5
                                for g in range(4):
                                    for e in range(4):
                                        acc[wave, lane, ci, pi, 4 * g + e] += Dm[8 * g + 4 * khalf + e, r31]

        # prologue of the workgroup's first tile
        i = wg
        q0, co0 = tile_of(i)
        xoff = make_xoff(q0)
        for k in range(SPW):
            issue_slab_piece(xoff, 0, 0, k)
        for d in range(D):
            issue_w(co0, 0, d, d)
        cur = read_frags(0, 0, 0)
        vbase = 0
        while True:
            nxt_tile = tile_of(i + stride) if G else None
            has_next = nxt_tile is not None
            acc = np.zeros((8, 64, 2, 2, 16), np.float32)     # [wave][lane][ci][pi][v]
            for cs in range(csteps):
                fin = cs + 1 >= csteps
                nomore, nxt = fin and not has_next, fin and has_next
                if nxt:
                    xoff = make_xoff(nxt_tile[0])              # refreshed before the last slice
                for tap in range(9):
                    s = 9 * (vbase + cs) + tap                 # global step number of this workgroup
                    st = s % NW
                    assert st == (tap % 3 if NW == 3 else (vbase + cs + tap) & 3)
                    carry = tap + D >= 9
                    if not nomore or not carry:
                        over = carry and nxt
                        issue_w(nxt_tile[1] if over else co0, 0 if over else cs + (1 if carry else 0), (tap + D) % 9, st)
                    if tap < SPW and not nomore:
                        issue_slab_piece(xoff, (cs + 1) & 1, 0 if nxt else (cs + 1) * 64, tap)
                    # NOTE: the emulation completes loads instantly, so it checks addressing, not the wait counts
                    last_step = nomore and tap == 8
                    nxt_frags = None if last_step else read_frags(cs + (1 if tap == 8 else 0), (tap + 1) % 9, (s + 1) % NW)
                    mfma_step(acc, *cur)
                    cur = nxt_frags
            # epilogue: two passes through the slab buffer of the last (odd) slice
            stage_lo = (SLAB0 + SLB) // 16
            lds[stage_lo:stage_lo + 32768 // 16] = np.nan          # the stage clobbers that buffer (and nothing else)
            for wave in range(8):
                wm, wn = wave >> 2, wave & 3
                if pool:
                    b, h0, w0 = tile_origin(q0)
                    pooled = np.full((64, 2, 16), np.nan, np.float32)           # [lane][ci][v] after the vertical + horizontal maxima
                    for lane in range(64):
                        r31 = lane & 31
                        slot2 = wn * 32 + r31
                        pair, col = slot2 >> csh, slot2 & (TC - 1)
                        has_below, has_right = h0 + 2 * pair + 1 < H, w0 + col + 1 < W
                        for ci in range(2):
                            v = acc[wave, lane, ci, 0].copy()
                            if has_below:
                                v = np.maximum(v, acc[wave, lane, ci, 1])
                            pooled[lane, ci] = v
                    hp = pooled.copy()
                    for lane in range(64):
                        slot2 = wn * 32 + (lane & 31)
                        if w0 + (slot2 & (TC - 1)) + 1 < W:
                            hp[lane] = np.maximum(pooled[lane], pooled[lane ^ 1])      # DPP quad_perm [1, 0, 3, 2]
                    stage = np.full((16, 8, 8), np.nan, np.float32)
                    for lane in range(64):
                        r31, khalf = lane & 31, lane >> 5
                        if r31 & 1:
                            continue
                        px = r31 >> 1
                        for ci in range(2):
                            for g in range(4):
                                stage[px, (ci * 4 + g) ^ (px & 7), khalf * 4:khalf * 4 + 4] = hp[lane, ci, 4 * g:4 * g + 4]
                    for lane in range(64):
                        for j in range(2):
                            idx = j * 64 + lane
                            px, c = idx >> 3, idx & 7
                            se = wn * 32 + 2 * px
                            ho, wo = (h0 >> 1) + (se >> csh), (w0 + (se & (TC - 1))) >> 1
                            if ho < Ho and wo < Wo:
                                ch = co0 + wm * 64 + c * 8
                                assert np.isnan(y[b, ho, wo, ch])
                                y[b, ho, wo, ch:ch + 8] = stage[px, c ^ (px & 7)]
                    continue
                for pi in range(2):
                    stage = np.full((32, 8, 8), np.nan, np.float32)            # [px][chunk position][8 channels]
                    for lane in range(64):
                        r31, khalf = lane & 31, lane >> 5
                        for ci in range(2):
                            for g in range(4):
                                chunk = ci * 4 + g
                                stage[r31, chunk ^ (r31 & 7), khalf * 4:khalf * 4 + 4] = acc[wave, lane, ci, pi, 4 * g:4 * g + 4]
                    for lane in range(64):
                        c = lane & 7
                        if G2:
                            b, h0, w0 = tile_origin(q0)
                            for j in range(4):
                                px = j * 8 + (lane >> 3)
                                sl2 = wn * 32 + px
                                hh, ww = h0 + 2 * (sl2 >> csh) + pi, w0 + (sl2 & (TC - 1))
                                if hh < H and ww < W:
                                    ch = co0 + wm * 64 + c * 8
                                    assert np.isnan(y[b, hh, ww, ch])
                                    y[b, hh, ww, ch:ch + 8] = stage[px, c ^ (px & 7)]
                            continue
                        q = q0 + wn * 64 + pi * 32 + (lane >> 3)
                        b = q // (H1 * W1); r = q - b * H1 * W1; h = r // W1; w = r - h * W1
                        for j in range(4):
                            px = j * 8 + (lane >> 3)
                            v = stage[px, c ^ (px & 7)]
                            if w < W and h < H and q < Q:
                                ch = co0 + wm * 64 + c * 8
                                assert np.isnan(y[b, h, w, ch])
                                y[b, h, w, ch:ch + 8] = v
                            q += 8; w += 8
                            while w >= W1:
                                w -= W1; h += 1
                                if h == H1:
                                    h = 0; b += 1
            if not has_next:
                break
            i += stride
            q0, co0 = nxt_tile
            vbase += csteps

    # direct convolution
    xp = np.zeros((B, H + 2, W + 2, Cin), np.float32)
    xp[:, 1:-1, 1:-1] = x
    ref = np.zeros((B, H, W, Cout), np.float32)
    for kh in range(3):
        for kw in range(3):
            ref += np.einsum("bhwc,oc->bhwo", xp[:, kh:kh + H, kw:kw + W], wt[:, kh, kw])
    if pool:                                              # MaxPooling2D(2, 2, 'same'): windows clipped to the map
        pr = np.full((B, Ho, Wo, Cout), -np.inf, np.float32)
        for dh in range(2):
            for dw in range(2):
                part = ref[:, dh::2, dw::2]
                pr[:, :part.shape[1], :part.shape[2]] = np.maximum(pr[:, :part.shape[1], :part.shape[2]], part)
        ref = pr
    return np.array_equal(y, ref), y, ref


if __name__ == "__main__":
    ok = True
    cases = [(2, 5, 6, 128, 128, 4, 5, None, 0, False), (1, 9, 40, 128, 128, 4, 6, None, 0, False), (3, 3, 70, 128, 128, 3, 7, None, 0, False),
             (2, 19, 19, 256, 256, 4, 5, None, 0, False),
             (24, 19, 19, 128, 256, 4, 5, 8, 0, False), (5, 38, 38, 128, 128, 4, 6, 8, 0, False), (3, 40, 75, 128, 256, 3, 7, 8, 0, False),
             (40, 7, 9, 256, 128, 4, 5, 8, 0, False),
             (2, 20, 37, 128, 128, 4, 6, None, 4, True), (1, 9, 40, 128, 128, 4, 6, 8, 5, False), (3, 17, 33, 128, 256, 4, 6, 8, 5, True),
             (2, 33, 18, 128, 128, 4, 6, 8, 4, False)]
    if len(sys.argv) > 1 and sys.argv[1] == "2d":
        cases = cases[8:]
    for (B, H, W, Cin, Cout, NW, SPW, G, csh, pool) in cases:
        good, y, ref = emulate(B, H, W, Cin, Cout, NW, SPW, G=G, csh=csh, pool=pool)
        print((B, H, W, Cin, Cout, NW, SPW, G, csh, pool), "OK" if good else "MISMATCH (%d wrong, %d unwritten)" % ((y != ref).sum(), np.isnan(y).sum()))
        ok &= good
    sys.exit(0 if ok else 1)
