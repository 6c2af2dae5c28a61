"""Lane-level emulation of conv3x3_kernel / conv3x3_stacked_kernel (go_slam_amd/csrc/conv3x3.hip) on the CPU: every
thread's index arithmetic -- patch staging, weight images, fragment addresses, the MFMA operand / accumulator register
layout of v_mfma_f32_32x32x16_f16, the per-pixel masking of the row-stacked tiling, the LDS-transposed epilogue -- is
replayed in NumPy and the result compared with F.conv2d.  It is how the kernel variants were checked before they ever
ran on a GPU (both hardware-verified variants passed first time), and how the lane-permuted (LP) instantiations are
checked until they do.  `bank_model()` evaluates the ds_read_b128 service groups of MI355X_MICROARCH.md (LDS) for the
pixel-fragment reads.  This file restates the kernel by hand: keep it in step with conv3x3.hip.

    python tools/emulate_conv3x3.py          # bank model + a set of shapes (minutes)
"""
import numpy as np, torch, torch.nn.functional as F, sys, os
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd.droid_net import pack_conv3x3_weight
BN=128; TS=72
def frag_lane(r):
    q=r>>2; return (0x96>>q)&1, r-4*((q+1)>>1)
def tile_pixel(TW, LP, wm, i, r):
    if (not LP) or TW==32:
        m=wm*128+i*32+r; return m//TW, m%TW
    grp,pos=frag_lane(r)
    if TW==16: return wm*8+2*i+grp, pos
    return wm*16+i+8*grp+4*(pos>>3), pos&7
def run(x, xs, C, wpack, O, n, H, W, KC, TW, LP, stacked, xb=None, xsb=0, split=0, epi=None):
    TH=256//TW; PW=TW+2; NP=(TH+2)*PW; NPP=NP+1 if LP else NP; KG=KC//8; WTAP=KG*BN
    rows=n*H
    if stacked:
        tiles_x=(W+TW-1)//TW; tiles_y=(rows+TH-1)//TH; nblk=tiles_x*tiles_y
    else:
        tiles_x=(W+TW-1)//TW; tiles_y=(H+TH-1)//TH; nblk=n*tiles_x*tiles_y
    y=np.full((rows*W, O), np.nan, np.float32)
    wp=wpack.reshape(-1,8)
    for bx in range(nblk):
      for nb in range(O//BN):
        if stacked:
            tx0=(bx%tiles_x)*TW; g0=(bx//tiles_x)*TH
        else:
            t=bx; tx0=(t%tiles_x)*TW; t//=tiles_x; ty0=(t%tiles_y)*TH; img=t//tiles_y
        nchunk=C//KC; wsrc=nb*nchunk*9*WTAP
        acc=np.zeros((4,64,2,4,16),np.float32)
        patch=np.zeros((KG*NPP,8),np.float32)
        for ck in range(nchunk):
            for it in range(KG*NP):
                kg=it&(KG-1); p=it//KG; pr=p//PW; pc=p-pr*PW
                v=np.zeros(8,np.float32)
                if stacked:
                    gv=g0+pr-1; gx=tx0+pc-1
                    if 0<=gv<rows and 0<=gx<W:
                        base=(gv*W+gx)*xs+ck*KC+kg*8; v=x[base:base+8]
                else:
                    gy=ty0+pr-1; gx=tx0+pc-1
                    if 0<=gy<H and 0<=gx<W:
                        pix=(img*H+gy)*W+gx; c0=ck*KC+kg*8
                        if xb is not None and c0>=split:        # EPI 2: second input source
                            base=pix*xsb+(c0-split); v=xb[base:base+8]
                        else:
                            base=pix*xs+c0; v=x[base:base+8]
                patch[kg*NPP+p]=v
            wck=wsrc+ck*9*WTAP
            for tap in range(9):
                wb=wp[wck+tap*WTAP: wck+(tap+1)*WTAP]; dy=tap//3; toff=dy*PW+(tap%3)
                for wv in range(4):
                    wm=wv&1; wn=wv>>1
                    for grp in range(KC//32):
                      for s in range(2):
                        for j in range(2):
                            A=np.zeros((32,16),np.float32)
                            for lane in range(64):
                                r=lane&31; kgl=lane>>5; kg=4*grp+2*s+kgl
                                A[r,8*kgl:8*kgl+8]=wb[kg*BN+wn*64+j*32+r]
                            for i in range(4):
                                B=np.zeros((32,16),np.float32)
                                for lane in range(64):
                                    r=lane&31; kgl=lane>>5; kg=4*grp+2*s+kgl
                                    ty,tx=tile_pixel(TW,LP,wm,i,r)
                                    v=patch[kg*NPP+ty*PW+tx+toff]
                                    if stacked:
                                        yy=(g0+ty)%H
                                        if dy==0 and yy==0: v=np.zeros(8)
                                        if dy==2 and yy==H-1: v=np.zeros(8)
                                    B[r,8*kgl:8*kgl+8]=v
                                Cm=A@B.T
                                for lane in range(64):
                                    col=lane&31
                                    for reg in range(16):
                                        row=(reg&3)+8*(reg>>2)+4*(lane>>5)
                                        acc[wv,lane,j,i,reg]+=Cm[row,col]
        for wv in range(4):
            wm=wv&1; wn=wv>>1
            for i in range(4):
                tile=np.zeros((32*TS,),np.float32)
                for lane in range(64):
                    r=lane&31; kgl=lane>>5
                    for j in range(2):
                        for g in range(4):
                            for e in range(4):
                                tile[r*TS+j*32+8*g+4*kgl+e]=acc[wv,lane,j,i,4*g+e]
                for it in range(4):
                    for lane in range(64):
                        pxr=it*8+(lane>>3); piece=lane&7
                        ty,tx=tile_pixel(TW,LP,wm,i,pxr)
                        if stacked:
                            gv=g0+ty; gx=tx0+tx; ok=gv<rows and gx<W
                        else:
                            gy=ty0+ty; gx=tx0+tx; ok=gy<H and gx<W; gv=img*H+gy
                        if ok:
                            c0=nb*BN+wn*64+piece*8
                            v8=tile[pxr*TS+piece*8:pxr*TS+piece*8+8]
                            if epi is None:                      # kept in fp32 here so that index errors cannot hide
                                y[gv*W+gx, c0:c0+8]=v8           # behind the fp16 rounding of the real LDS tile
                            else:
                                epi(gv*W+gx, gv//H, nb, wn*64+piece*8, v8.astype(np.float16).astype(np.float32))
    return y


def decode_block(L, ntiles, NB, xcd):
    """conv3x3.hip decode_block: workgroup id -> (tile, output-channel block)"""
    if not xcd:
        return L % ntiles, L // ntiles
    G = 8 * NB
    s, l = L // G, L % G
    m = min(8, ntiles - s * 8)
    return s * 8 + l % m, l // m


def gru_fused(n, H, W, c_rest, LP, hoisted=True, seed=3):
    """Emulates gs_conv3x3_gru_zr + gs_conv3x3_gru_q (EPI 1 / 2 of conv3x3_kernel) on random data and returns the max
    difference to conv (fp16-rounded) + the gate formulas of gru_gates.hip evaluated densely in torch, for z, r * net
    and the new hidden state, plus the count of outputs never written."""
    g = torch.Generator().manual_seed(seed)
    cin = 128 + c_rest
    P = n * H * W
    hx = (torch.randn(P, cin, generator=g) * 0.5).half()
    net = hx[:, :128].clone()
    wzr = (torch.randn(256, cin, 3, 3, generator=g) * 0.03).half()
    wq = (torch.randn(128, cin, 3, 3, generator=g) * 0.03).half()
    bzr, bq = torch.randn(256, generator=g) * 0.1, torch.randn(128, generator=g) * 0.1
    gzr, gq = torch.randn(n, 256, generator=g) * 0.1, torch.randn(n, 128, generator=g) * 0.1
    inp_pre = (torch.randn(P, 384, generator=g) * 0.3).half() if hoisted else None
    f32 = lambda t: t.float().numpy()
    sig = lambda v: 1.0 / (1.0 + np.exp(-v, dtype=np.float32))
    z_o = np.full((P, 128), np.nan, np.float32); rn_o = z_o.copy(); out_o = z_o.copy()
    ip = f32(inp_pre) if hoisted else None

    def epi1(pix, img, nb, c8, v8):
        pi = ip[pix, nb * 128 + c8: nb * 128 + c8 + 8] if hoisted else np.zeros(8, np.float32)
        a = sig((v8 + pi + f32(bzr)[nb * 128 + c8: nb * 128 + c8 + 8] + f32(gzr)[img, nb * 128 + c8: nb * 128 + c8 + 8]).astype(np.float32))
        if nb == 0:
            z_o[pix, c8:c8 + 8] = a.astype(np.float16)
        else:
            rn_o[pix, c8:c8 + 8] = (a * f32(hx)[pix, c8:c8 + 8]).astype(np.float16)
    run(f32(hx).reshape(-1), cin, cin, pack_conv3x3_weight(wzr, 32).float().numpy(), 256, n, H, W, 32, 16, LP, False, epi=epi1)
    z16, rn16 = z_o.copy(), rn_o.copy()

    def epi2(pix, img, nb, c8, v8):
        pi = ip[pix, 256 + c8: 256 + c8 + 8] if hoisted else np.zeros(8, np.float32)
        a = (v8 + pi + f32(bq)[c8:c8 + 8] + f32(gq)[img, c8:c8 + 8]).astype(np.float32)
        q = (1.0 - 2.0 / (1.0 + np.exp(2.0 * a, dtype=np.float32))).astype(np.float32)
        zf = z16[pix, c8:c8 + 8]
        out_o[pix, c8:c8 + 8] = ((1.0 - zf) * f32(net)[pix, c8:c8 + 8] + zf * q).astype(np.float16)
    rest = hx[:, 128:].contiguous()
    run(rn16.reshape(-1), 128, cin, pack_conv3x3_weight(wq, 64).float().numpy(), 128, n, H, W, 64, 16, LP, False,
        xb=f32(rest).reshape(-1), xsb=c_rest, split=128, epi=epi2)
    # dense reference: conv in fp32 on the fp16 operands, rounded to fp16, then the gate formulas
    to_img = lambda t: t.float().view(n, H, W, -1).permute(0, 3, 1, 2)
    zr_pre = F.conv2d(to_img(hx), wzr.float(), padding=1).permute(0, 2, 3, 1).reshape(P, 256).half().float()
    addz = zr_pre + (inp_pre[:, :256].float() if hoisted else 0) + bzr + gzr.repeat_interleave(H * W, 0)
    zr = torch.sigmoid(addz)
    z_ref = zr[:, :128].half()
    rn_ref = (zr[:, 128:] * net.float()).half()
    hq = torch.cat([rn_ref, hx[:, 128:]], 1)
    q_pre = F.conv2d(to_img(hq), wq.float(), padding=1).permute(0, 2, 3, 1).reshape(P, 128).half().float()
    a = q_pre + (inp_pre[:, 256:].float() if hoisted else 0) + bq + gq.repeat_interleave(H * W, 0)
    out_ref = ((1 - z_ref.float()) * net.float() + z_ref.float() * torch.tanh(a)).half()
    d = lambda e, r: float(np.abs(np.nan_to_num(e) - r.float().numpy()).max())
    unwritten = int(np.isnan(z_o).sum() + np.isnan(rn_o).sum() + np.isnan(out_o).sum())
    return d(z_o, z_ref), d(rn_o, rn_ref), d(out_o, out_ref), unwritten


def bank_model():
    """worst n-way conflict of a B-fragment ds_read_b128 per service group, for each tile width / lane mapping"""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    out = {}
    for TW in (16, 8, 32):
        PW = TW + 2
        for LP in (False, True):
            worst = 0
            for wm in range(2):
                for i in range(4):
                    for g in groups:
                        c = Counter((tile_pixel(TW, LP, wm, i, r)[0] * PW + tile_pixel(TW, LP, wm, i, r)[1]) % 16 for r in g)
                        worst = max(worst, max(c.values()))
            out[(TW, LP)] = worst
    return out


def check(n, H, W, C, O, xs, KC, TW, LP, stacked, seed=1):
    """max |emulated kernel - F.conv2d| and the number of output elements the emulated kernel never wrote"""
    g = torch.Generator().manual_seed(seed)
    xt = torch.randn(n, H, W, xs, generator=g).half()
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.1).half()
    wp = pack_conv3x3_weight(w, KC).float().numpy()
    y = run(xt.float().numpy().reshape(-1), xs, C, wp, O, n, H, W, KC, TW, LP, stacked)
    ref = F.conv2d(xt[..., :C].permute(0, 3, 1, 2).float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, O).numpy()
    return float(np.abs(np.nan_to_num(y) - ref).max()), int(np.isnan(y).sum())


if __name__ == "__main__":
    print("worst n-way conflict of a pixel-fragment read {(tile width, lane-permuted): n}:", bank_model())
    for TW in (8, 16, 32):                                   # every pixel of the tile is owned by exactly one lane
        seen = Counter(tile_pixel(TW, True, wm, i, r) for wm in range(2) for i in range(4) for r in range(32))
        assert len(seen) == 256 and set(seen.values()) == {1} and max(t for t, _ in seen) == 256 // TW - 1, TW
    for case in [(2, 20, 19, 32, 128, 40, 32, 16, False, False), (2, 20, 19, 32, 128, 40, 32, 16, True, False),
                 (1, 9, 17, 64, 128, 64, 64, 16, True, False), (3, 7, 13, 32, 128, 32, 32, 16, False, True),
                 (3, 7, 13, 32, 128, 32, 32, 16, True, True), (3, 7, 13, 32, 128, 40, 32, 8, True, True),
                 (2, 5, 37, 32, 128, 32, 32, 32, False, True), (2, 9, 10, 64, 128, 64, 64, 8, True, True)]:
        print(dict(zip("n H W C O xs KC TW LP stacked".split(), case)), "-> max diff, unwritten:", check(*case))
