"""Index arithmetic of go_slam_amd/csrc/corr_build.hip restated in NumPy (developer / CPU-test tool, not the product):
the fragment-ordered operand image corr_prep_kernel writes and corr_volume_kernel reads, the XCD-consecutive tile
renumbering, the tile8 plane addresses of levels 0 / 1 for 4-row tiles, and the whole pipeline (fragment-ordered
operands -> MFMA lane mapping -> LDS tile -> level stores -> pools) on small maps against a plain matmul + avg_pool."""
import numpy as np

KDIM, BM, ROWS = 128, 64, 4


def padded_pixels(h, w):
    a = (h * w + BM - 1) // BM * BM
    b = (h + ROWS - 1) // ROWS * ROWS * w
    return (max(a, b) + 31) // 32 * 32 + (32 if (ROWS * w) % 32 else 0)


def frag_offset(p, c):
    """element offset of (pixel p, channel c) in the fragment-ordered image: [tile][ks][lane = 32 hi + r][8]"""
    tile, r = p >> 5, p & 31
    ks, hi, k = c >> 4, (c >> 3) & 1, c & 7
    return (((tile * 8 + ks) * 64) + 32 * hi + r) * 8 + k


def prep(fmap, h, w):
    """[128, hw] -> fragment order, x 1/4, zero padded (corr_prep_kernel)"""
    hw = h * w
    P = padded_pixels(h, w)
    out = np.zeros(P * KDIM, dtype=np.float16)
    p, c = np.meshgrid(np.arange(hw), np.arange(KDIM), indexing="ij")
    out[frag_offset(p, c)] = (fmap.T.astype(np.float16) / np.float16(4.0)).astype(np.float16)
    return out


def fragment(img, tile, ks):
    """what `ld8(A + ((tile * 8 + ks) * 64 + lane) * 8)` gives the 64 lanes: [64, 8]"""
    base = (tile * 8 + ks) * 64 * 8
    return img[base:base + 512].reshape(64, 8)


def fragment_at(img, q0, nt, ks):
    """the target-pixel fragment of corr_volume_kernel's `frag(nt, ks)`: lane (hi, r) reads pixel q = q0 + 32 nt + r at
    ((q >> 5) * 8 + ks) * 64 + 32 hi + (q & 31) -- one stored tile when q0 % 32 == 0, two half tiles when q0 % 32 == 16"""
    out = np.zeros((64, 8), img.dtype)
    for lane in range(64):
        q = q0 + 32 * nt + (lane & 31)
        base = (((q >> 5) * 8 + ks) * 64 + (lane & 32) + (q & 31)) * 8
        out[lane] = img[base:base + 8]
    return out


def mfma_32x32x16(a, b, acc):
    """v_mfma_f32_32x32x16_f16: a, b [64 lanes, 8] (lane = 32 hi + row/col, k = 8 hi + e); acc [64 lanes, 16]
    with D[row][col]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)"""
    A = np.zeros((32, 16), np.float32)
    B = np.zeros((16, 32), np.float32)
    for lane in range(64):
        A[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = a[lane].astype(np.float32)
        B[8 * (lane >> 5):8 * (lane >> 5) + 8, lane & 31] = b[lane].astype(np.float32)
    D = A @ B
    for lane in range(64):
        for reg in range(16):
            acc[lane, reg] += D[(reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), lane & 31]
    return acc


def xcd_tile(block, nblocks):
    """tile index of launch index `block` (nblocks = the padded grid, a multiple of 8)"""
    return (block & 7) * (nblocks >> 3) + (block >> 3)


def tile8_addr(y, x, w):
    """element offset of (y, x) in a tile8 plane of width w"""
    return ((y >> 3) * (w >> 3) + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7)


def volume_tile(f1img, f2img, h, w, p1_0, y2_0):
    """one workgroup of corr_volume_kernel: returns the [BM, ROWS * w] fp16 tile c0 (source pixel m, target column j)"""
    BN = ROWS * w
    ntile = (BN + 31) // 32
    q0 = y2_0 * w
    c0 = np.zeros((BM, 32 * ntile), np.float16)          # (LD0 = 32 ntile + 8; columns >= BN are never stored)
    for nt in range(ntile):
        acc = [np.zeros((64, 16), np.float32) for _ in range(BM // 32)]
        for ks in range(8):
            af = fragment_at(f2img, q0, nt, ks)
            for mt in range(BM // 32):
                bfr = fragment(f1img, (p1_0 >> 5) + mt, ks)
                acc[mt] = mfma_32x32x16(af, bfr, acc[mt])
        for mt in range(BM // 32):
            for lane in range(64):
                r = lane & 31
                for q in range(4):
                    for k in range(4):
                        c0[32 * mt + r, 32 * nt + 8 * q + 4 * (lane >> 5) + k] = np.float16(acc[mt][lane, q * 4 + k])
    return c0[:, :BN]
