"""Target for the PMC passes over the implicit-GEMM 3x3 convolution: runs ONE layer shape a few times so that
`rocprofv3 --pmc ...` attributes the counters to conv3x3_kernel alone.

    cd /tmp && export TMPDIR=/tmp                     # as the environment notes ask
    rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/conv_trace -- python $GRAFT_REPO_ROOT/tools/profile_conv3x3.py
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \\
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d .../conv_pmc_sq -- python .../tools/profile_conv3x3.py
    rocprofv3 --pmc FETCH_SIZE -d .../conv_pmc_fetch -- python .../tools/profile_conv3x3.py     # separate passes:
    rocprofv3 --pmc WRITE_SIZE -d .../conv_pmc_write -- python .../tools/profile_conv3x3.py     # TCC slots
(counters in their own runs, never together with --sys-trace / --runtime-trace; summaries go to profiles/.)

    python tools/profile_conv3x3.py [layer] [kc] [stacked]     # layer in gru_zr (default), gru_q, heads, corr_enc2
Variants are selected by the environment (GOSLAM_CONV3X3_LANEPERM / _XCD), as in tools/conv3x3_variants.sh.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import droid_net as DN  # noqa: E402

LAYERS = {"gru_zr": (320, 256), "gru_q": (320, 128), "heads": (128, 384), "corr_enc2": (128, 128)}


def main():
    layer = sys.argv[1] if len(sys.argv) > 1 else "gru_zr"
    kc = int(sys.argv[2]) if len(sys.argv) > 2 else None
    stacked = len(sys.argv) > 3 and sys.argv[3] == "stacked"
    c, o = LAYERS[layer]
    dev = "cuda:0"
    x = torch.randn(75, c, 60, 80, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(o, c, 3, 3, device=dev) / (3 * c ** 0.5)).half()
    for _ in range(8):
        DN.conv3x3_hip(x, w, kc, stacked=stacked)
    torch.cuda.synchronize()
    flops = 2.0 * 75 * 60 * 80 * 9 * c * o
    print(f"{layer}: {flops / 1e9:.1f} GFLOP per launch; algorithmic bytes in {x.numel() * 2 / 1e6:.1f} MB + out "
          f"{75 * 60 * 80 * o * 2 / 1e6:.1f} MB + weights {w.numel() * 2 / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
