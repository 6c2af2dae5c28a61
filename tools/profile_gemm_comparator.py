"""The plain-GEMM comparator of conv3x3_pp: a library fp16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) of
the update operator's implicit-GEMM shapes on random data, as a stand-alone command for rocprofv3 --pmc / --kernel-trace
passes and with its own event timing:
    python tools/profile_gemm_comparator.py [gru_zr|gru_q|square] [launches]
gru_zr: M x N x K = 360000 x 256 x 2880 (75 edges x 60 x 80 pixels, 9 taps x 320 channels -> z|r), gru_q: x 128,
square: 8192^3 (the guide's reference GEMM form).  Prints one JSON line (median us, TFLOP/s, fraction of the 2.5 PF peak).
Whether the PART sustains ~1.9 GHz / >= 1.15 PF on a dense fp16 MFMA body of this shape ON THIS BOX -- i.e. whether
conv3x3_pp's 0.40 is the part or the kernel body -- is read off the same counters as tools/profile_conv3x3.py
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE / wall)."""
import json
import sys

import torch

SHAPES = {"gru_zr": (360000, 256, 2880), "gru_q": (360000, 128, 2880), "square": (8192, 8192, 8192)}
name = sys.argv[1] if len(sys.argv) > 1 else "gru_zr"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M, N, K = SHAPES[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).half()
b = (torch.randn(N, K, device=dev) / K ** 0.5).half()          # B^T layout ("NT": both operands K-contiguous, as the conv reads them)
out = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(3):
    torch.matmul(a, b.t(), out=out)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
for s, e in ev:
    s.record()
    torch.matmul(a, b.t(), out=out)
    e.record()
torch.cuda.synchronize()
us = sorted(1e3 * s.elapsed_time(e) for s, e in ev)
med = us[len(us) // 2]
tf = 2.0 * M * N * K / (med * 1e-6) / 1e12
print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "median_us": med, "min_us": us[0], "tflops": tf, "frac_of_2500TF": tf / 2500.0,
                  "launches": launches, "finite": bool(torch.isfinite(out.float()).all())}))
