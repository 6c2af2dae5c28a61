"""CPU emulation of conv3x3_pp_kernel (go_slam_amd/csrc/conv3x3_pp.hip), three independent checks:

  schedule()   replays the instruction stream of both wave groups on the barrier-interval timeline: every LDS-DMA
               issue, counted wait, barrier and fragment read, and checks for each read that the buffer holds the
               expected tap / chunk, that BOTH groups' pieces were retired by a wait that is followed by a barrier
               before the reading phase, and that no buffer is re-filled before its last reader is done.
  data_path()  replays every thread's index arithmetic in NumPy -- DMA source addresses incl. the zero page and the pad
               slot of the 80-byte pixel stride, linear LDS destinations, fragment slot addresses (per-lane base +
               immediate tap offset) incl. the zero region of the row-stacked masks,
               the operand / accumulator layout of v_mfma_f32_32x32x16_f16, the lane-permuted pixel mapping and the
               LDS-transposed epilogue -- and compares the result with F.conv2d.
  bank_model() evaluates the ds_read_b128 service groups (MI355X_MICROARCH.md, LDS) for the B-fragment reads.

This file restates the kernel by hand: keep it in step with conv3x3_pp.hip.      python tools/emulate_conv3x3_pp.py
"""
import os
import sys
from collections import Counter

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd.droid_net import pack_conv3x3_weight  # noqa: E402

PSTR = 5                                # 16-byte slots per patch pixel (80-byte stride)
BN, KG, WTAP, TS = 128, 4, 512, 72      # the 128-channel instantiation (schedule / bank model); data_path derives its own


def frag_lane(r):
    q = r >> 2
    return (0x96 >> q) & 1, r - 4 * ((q + 1) >> 1)


def tile_pixel(TW, wm, i, r):
    grp, pos = frag_lane(r)
    if TW == 16:
        return wm * 8 + 2 * i + grp, pos
    return wm * 16 + i + 8 * grp + 4 * (pos >> 3), pos & 7


def decode_block(L, ntiles, NB, xcd):
    """conv3x3_common.h decode_block: workgroup id -> (tile, output-channel block)"""
    if not xcd:
        return L % ntiles, L // ntiles
    G = 8 * NB
    s, l = L // G, L % G
    m = min(8, ntiles - s * 8)
    return s * 8 + l % m, l // m


def geometry(TW):
    TH = 512 // TW
    PW = TW + 2
    NPX = (TH + 2) * PW
    NROUND = (NPX * PSTR + 511) // 512
    zpad = (((2 * PW + 2) * PSTR + 3 + 15) // 16) * 16
    return TH, PW, NPX, NROUND, NROUND * 512 + zpad, NROUND * 512


# ---------------------------------------------------------------------------------------------------------------------
def schedule(nchunk, TW):
    """Timeline check.  Interval k = between a wave's k-th and (k+1)-th barrier."""
    _, _, _, NROUND, _, _ = geometry(TW)
    T = nchunk * 9
    events = {0: [], 1: []}          # group -> list of (interval, kind, buffer, tag)
    for g in (0, 1):
        k = 0                        # barriers executed so far
        pending = []                 # issued, not yet retired: (buffer, tag, issue_interval)
        retired = []                 # (buffer, tag, wait_interval)

        def issue(buf, tag):
            pending.append((buf, tag, k))
            events[g].append((k, "issue", buf, tag))

        def wait(n_left):            # s_waitcnt vmcnt(n_left): all but the newest n_left are complete
            nonlocal pending
            done, pending = pending[:len(pending) - n_left] if n_left else pending, pending[len(pending) - n_left:] if n_left else []
            for buf, tag, _ in done:
                retired.append((buf, tag, k))
                events[g].append((k, "retire", buf, tag))

        def read(buf, tag):
            events[g].append((k, "read", buf, tag))

        for q in range(NROUND):
            issue(("p", 0), ("chunk", 0))
        for t in range(3):
            issue(("w", t), ("tap", t))
        wait(0)
        k += 1                                           # barrier 0
        if g == 1:
            k += 1                                       # the extra barrier
        read(("p", 0), ("chunk", 0)); read(("w", 0), ("tap", 0))
        k += 1
        for ck in range(nchunk):
            for tap in range(9):
                tg = ck * 9 + tap
                issue(("w", (tg + 3) & 3), ("tap", min(tg + 3, T - 1)) if tg + 3 < T else ("dead", tg))
                if tap < NROUND:
                    issue(("p", (ck + 1) & 1), ("chunk", ck + 1) if ck + 1 < nchunk else ("dead", tg))
                # math phase: no LDS access
                wait(2 if tap < NROUND else 1)
                k += 1
                if tap < 8:
                    read(("p", ck & 1), ("chunk", ck)); read(("w", (tg + 1) & 3), ("tap", tg + 1))
                elif ck + 1 < nchunk:
                    read(("p", (ck + 1) & 1), ("chunk", ck + 1)); read(("w", (tg + 1) & 3), ("tap", tg + 1))
                k += 1
        if g == 0:
            k += 1
        wait(0)
        k += 1
        events[g].append((k, "end", None, None))
    assert events[0][-1][0] == events[1][-1][0], "barrier counts of the two groups differ"
    # merge and check
    bufs = {}
    for g in (0, 1):
        for (iv, kind, buf, tag) in events[g]:
            if buf is not None:
                bufs.setdefault(buf, []).append((iv, kind, g, tag))
    problems = []
    for buf, ev in bufs.items():
        ev.sort(key=lambda e: (e[0], {"read": 0, "retire": 1, "issue": 2}[e[1]]))
        tags = []                                         # distinct fills in issue order
        for iv, kind, g, tag in ev:
            if kind == "issue" and (not tags or tags[-1] != tag):
                tags.append(tag)
        for tag in tags:
            issues = [e for e in ev if e[1] == "issue" and e[3] == tag]
            retires = [e for e in ev if e[1] == "retire" and e[3] == tag]
            reads = [e for e in ev if e[1] == "read" and e[3] == tag]
            if tag[0] == "dead":
                if reads:
                    problems.append((buf, tag, "a dead fill is read"))
                continue
            assert len({e[2] for e in retires}) == 2, (buf, tag, "not retired by both groups")
            valid_from = max(e[0] for e in retires) + 1   # wait, THEN a barrier, then readable
            for e in reads:
                if e[0] < valid_from:
                    problems.append((buf, tag, f"group {e[2]} reads in interval {e[0]}, valid from {valid_from}"))
            # reads of the PREVIOUS content must be over before the first issue of this one
            first_issue = min(e[0] for e in issues)
            prev_reads = [e for e in ev if e[1] == "read" and e[3] != tag and e[0] >= 0 and
                          tags.index(e[3]) < tags.index(tag)] if tag in tags else []
            for e in prev_reads:
                if e[0] >= first_issue:
                    problems.append((buf, tag, f"issued in interval {first_issue} while {e[3]} is read in {e[0]}"))
        # every read must find the most recent non-dead fill before it
        for iv, kind, g, tag in ev:
            if kind == "read":
                fills = [e for e in ev if e[1] == "issue" and e[0] <= iv]
                last = fills[-1][3]
                newer = [e for e in ev if e[1] == "issue" and e[0] <= iv and tags.index(e[3]) > tags.index(tag)]
                if newer:
                    problems.append((buf, tag, f"read in {iv} after a newer fill {newer[0][3]} was issued in {newer[0][0]}"))
    return problems


# ---------------------------------------------------------------------------------------------------------------------
def bank_model(TW):
    TH, PW, NPX, _, _, _ = geometry(TW)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    worst = Counter()
    for wm in range(4):
        for i in range(4):
            for tap in range(9):
                toff = (tap // 3) * PW + tap % 3
                for kgl in (0, 1):
                    for s in (0, 1):
                        for grp in groups:
                            slots = []
                            for r in grp:
                                ty, tx = tile_pixel(TW, wm, i, r)
                                p = ty * PW + tx + toff
                                s0 = p * PSTR + kgl
                                slots.append((s0 + 2) if s else s0)
                            ways = max(Counter(sl % 16 for sl in slots).values())
                            worst[ways] += 1
    return dict(worst)


# ---------------------------------------------------------------------------------------------------------------------
def data_path(x, xs, C, wt, n, H, W, TW, xcd=0):
    """x: flat float32 array [n*H*W*xs] (fp16 values); wt torch [O,C,3,3] half.  Returns y [n*H*W, O] float32."""
    O = wt.shape[0]
    BN = 128 if O % 128 == 0 else 64          # pp_block_channels: the 64-channel instantiation for O % 128 != 0
    WTAP, NJ = KG * BN, BN // 64
    TS = 72 if NJ == 2 else 40
    PIECES = NJ * 4
    PXIT, NIT = 64 // PIECES, 32 // (64 // PIECES)
    TH, PW, NPX, NROUND, PSLOTS, ZSLOT = geometry(TW)
    wp = pack_conv3x3_weight(wt, 32).float().numpy().reshape(-1, 8)
    rows = n * H
    tiles_x = (W + TW - 1) // TW
    tiles_y = (rows + TH - 1) // TH
    NB = O // BN
    ntiles = tiles_x * tiles_y
    nchunk = C // 32
    T = nchunk * 9
    y = np.full((rows * W, O), np.nan, np.float32)
    tid = np.arange(512)
    for L in range(ntiles * NB):
        if xcd:
            G = 8 * NB
            s_, l_ = L // G, L % G
            m_ = min(8, ntiles - s_ * 8)
            tix, nb = s_ * 8 + l_ % m_, l_ // m_
        else:
            tix, nb = L % ntiles, L // ntiles
        tx0 = (tix % tiles_x) * TW
        g0 = (tix // tiles_x) * TH
        wsrc = nb * T * WTAP
        # DMA sources
        poff = np.full((NROUND, 512), -1, np.int64)
        for q in range(NROUND):
            s = q * 512 + tid
            p = s // PSTR
            kg = s - PSTR * p
            pr, pc = p // PW, p % PW
            gv, gx = g0 + pr - 1, tx0 + pc - 1
            ok = (p < NPX) & (kg < 4) & (gv >= 0) & (gv < rows) & (gx >= 0) & (gx < W)
            poff[q] = np.where(ok, (gv * W + gx) * xs + kg * 8, -1)
        pbuf = np.full((2, PSLOTS, 8), np.nan, np.float32)
        pbuf[:, ZSLOT:] = 0.0                            # the zero regions (written once, never a DMA target)
        wbuf = np.full((4, WTAP, 8), np.nan, np.float32)

        def dma_patch(chunk, buf):
            for q in range(NROUND):
                for t in range(512):
                    o = poff[q, t]
                    pbuf[buf, q * 512 + t] = x[o + chunk * 32: o + chunk * 32 + 8] if o >= 0 else 0.0

        def dma_w(tap_g, buf):
            wbuf[buf] = wp[wsrc + tap_g * WTAP: wsrc + (tap_g + 1) * WTAP]

        acc = np.zeros((8, 64, NJ, 4, 16), np.float32)
        dma_patch(0, 0)
        for t in range(3):
            dma_w(t, t)
        for ck in range(nchunk):
            if ck + 1 < nchunk:
                dma_patch(ck + 1, (ck + 1) & 1)          # (issued round by round during taps 0.. of chunk ck)
            for tap in range(9):
                tg = ck * 9 + tap
                # NOTE the ring: the fill for tap tg+3 targets buffer (tg+3)&3 while taps tg..tg+2 are resident
                wb = wbuf[tg & 3].copy()
                if tg + 3 < T:
                    dma_w(tg + 3, (tg + 3) & 3)
                pp = pbuf[ck & 1]
                dy = tap // 3
                toff = dy * PW + tap % 3
                for wv in range(8):
                    grp2 = wv >> 2
                    wm = (wv & 1) + 2 * grp2
                    wn = (wv >> 1) & 1
                    for s in range(2):
                        A = np.zeros((NJ, 32, 16), np.float32)
                        for lane in range(64):
                            r, kgl = lane & 31, lane >> 5
                            kg = 2 * s + kgl
                            for j in range(NJ):
                                A[j, r, 8 * kgl:8 * kgl + 8] = wb[kg * BN + wn * (BN // 2) + j * 32 + r]
                        for i in range(4):
                            B = np.zeros((32, 16), np.float32)
                            for lane in range(64):
                                r, kgl = lane & 31, lane >> 5
                                ty, tx = tile_pixel(TW, wm, i, r)
                                yy = (g0 + ty) % H
                                base = (ty * PW + tx) * PSTR + kgl
                                if (dy == 0 and yy == 0) or (dy == 2 and yy == H - 1):
                                    base = ZSLOT
                                s0 = base + toff * PSTR
                                s1 = s0 + 2
                                B[r, 8 * kgl:8 * kgl + 8] = pp[s1 if s else s0]
                            for j in range(NJ):
                                Cm = A[j] @ B.T
                                for lane in range(64):
                                    col = lane & 31
                                    for reg in range(16):
                                        row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
                                        acc[wv, lane, j, i, reg] += Cm[row, col]
        for wv in range(8):
            grp2 = wv >> 2
            wm = (wv & 1) + 2 * grp2
            wn = (wv >> 1) & 1
            for i in range(4):
                tile = np.zeros((32 * TS,), np.float32)
                for lane in range(64):
                    r, kgl = lane & 31, lane >> 5
                    for j in range(NJ):
                        for g in range(4):
                            for e in range(4):
                                tile[r * TS + j * 32 + 8 * g + 4 * kgl + e] = acc[wv, lane, j, i, 4 * g + e]
                for it in range(NIT):
                    for lane in range(64):
                        pxr, piece = it * PXIT + lane // PIECES, lane & (PIECES - 1)
                        ty, tx = tile_pixel(TW, wm, i, pxr)
                        gv, gx = g0 + ty, tx0 + tx
                        if gv < rows and gx < W:
                            c0 = nb * BN + wn * (BN // 2) + piece * 8
                            y[gv * W + gx, c0:c0 + 8] = tile[pxr * TS + piece * 8: pxr * TS + piece * 8 + 8]
    return y


def check(n, H, W, C, xs, O, TW, xcd=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, H, W, xs, generator=g).half()
    wt = (torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).half()
    y = data_path(x.float().numpy().reshape(-1), xs, C, wt, n, H, W, TW, xcd)
    ref = F.conv2d(x[..., :C].permute(0, 3, 1, 2).float(), wt.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, O)
    nan = int(np.isnan(y).sum())
    err = float(np.nanmax(np.abs(y - ref.numpy())))
    return nan, err


if __name__ == "__main__":
    for TW in (16, 8):
        print("bank model TW", TW, bank_model(TW))
        for nchunk in (1, 2, 4, 10):
            pr = schedule(nchunk, TW)
            print(f"schedule TW={TW} nchunk={nchunk}:", "OK" if not pr else pr[:6])
    for args in ((2, 5, 19, 32, 40, 128, 16, 0), (3, 7, 10, 64, 64, 128, 8, 1), (1, 33, 16, 32, 32, 256, 16, 1),
                 (2, 9, 21, 64, 64, 64, 16, 1)):
        print("data path", args, check(*args))
