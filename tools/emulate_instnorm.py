"""The statistics scheme of go_slam_amd/csrc/instnorm.hip restated in NumPy fp32 (developer / CPU-test tool): per
256-pixel chunk the sums of d = x - k and d^2 with k = the chunk's first pixel, accumulated per thread (4 samples in
flight, `lanes` threads per channel group) and added through LDS; chunk moments (n, mean, M2); Chan merges of the chunks
in the order instnorm_final_kernel uses (per channel `per` threads take every per-th chunk, 8 at a time, then the
threads' results are merged in thread order); and the apply kernel's rounding chain."""
import numpy as np

IN_CHUNK = 256
f32 = np.float32


def merge(a, b):
    """Chan et al. on (n, mean, M2) triples, fp32, as `merge` in instnorm.hip"""
    if b[0] == 0:
        return a
    if a[0] == 0:
        return b
    n = f32(a[0] + b[0])
    d = f32(b[1] - a[1])
    f = f32(b[0] / n)
    return (n, f32(a[1] + d * f), f32(f32(a[2] + b[2]) + f32(f32(d * d) * a[0]) * f))


def chunk_moments(x, c):
    """x [pixels, c] fp16 (bias already added): (n, mean, M2) per channel of one chunk, thread by thread"""
    c8n = c // 8
    lanes = 256 // c8n
    k = x[0].astype(f32)
    s1 = np.zeros((lanes, c), f32)
    s2 = np.zeros((lanes, c), f32)
    for pl in range(lanes):
        for p in range(pl, x.shape[0], lanes):
            d = x[p].astype(f32) - k
            s1[pl] = s1[pl] + d
            s2[pl] = (d * d + s2[pl]).astype(f32)           # fmaf: one rounding
    a1 = np.zeros(c, f32)
    a2 = np.zeros(c, f32)
    for pl in range(lanes):                                  # the LDS column sums, lane after lane
        a1 = a1 + s1[pl]
        a2 = a2 + s2[pl]
    n = f32(x.shape[0])
    mean = k + a1 / n
    m2 = np.maximum(a2 - a1 * a1 / n, f32(0))
    return n, mean.astype(f32), m2.astype(f32)


def image_stats(x, eps=1e-5):
    """x [hw, c] fp16 -> (mean, invstd) fp32 per channel, as instnorm_stats_kernel + instnorm_final_kernel"""
    hw, c = x.shape
    chunks = [chunk_moments(x[p0:p0 + IN_CHUNK], c) for p0 in range(0, hw, IN_CHUNK)]
    nblk = len(chunks)
    per = 256 // min(c, 256)
    mean = np.zeros(c, f32)
    invstd = np.zeros(c, f32)
    for ch in range(c):
        parts = []
        for part in range(per):
            acc = (f32(0), f32(0), f32(0))
            for b in range(part, nblk, per):
                n, m, m2 = chunks[b]
                acc = merge(acc, (n, m[ch], m2[ch]))
            parts.append(acc)
        r = parts[0]
        for q in range(1, per):
            r = merge(r, parts[q])
        var = f32(r[2] / r[0]) if r[0] > 0 else f32(0)
        mean[ch] = r[1]
        invstd[ch] = f32(1.0) / np.sqrt(f32(var + f32(eps)), dtype=f32)
    return mean, invstd


def norm_act(x, bias, skip, instance, relu_in, relu_out, eps=1e-5):
    """the whole gs_norm_act on one image: x, skip [hw, c] fp16, bias [c] fp16 or None"""
    v = x
    if bias is not None:
        v = (x.astype(f32) + bias.astype(f32)).astype(np.float16)
    if instance:
        mean, invstd = image_stats(v, eps)
        v = ((v.astype(f32) - mean) * invstd).astype(np.float16)
    if relu_in:
        v = np.maximum(v, np.float16(0))
    if skip is not None:
        v = (skip.astype(f32) + v.astype(f32)).astype(np.float16)
    if relu_out:
        v = np.maximum(v, np.float16(0))
    return v
