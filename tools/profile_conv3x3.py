"""One layer shape of the update operator's 3x3 convolution (gs_conv3x3_pp) as a stand-alone command for rocprofv3
(--pmc passes through tools/pmc_pass.sh, or --kernel-trace):  python tools/profile_conv3x3.py [gru_zr|gru_q|heads|corr_enc2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import droid_net as DN  # noqa: E402

LAYERS = {"gru_zr": (320, 256), "gru_q": (320, 128), "heads": (128, 384), "corr_enc2": (128, 128)}
c, o = LAYERS[sys.argv[1] if len(sys.argv) > 1 else "gru_zr"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(75, c, 60, 80, device=dev).half().contiguous(memory_format=torch.channels_last)
w = (torch.randn(o, c, 3, 3, device=dev) / (3.0 * c ** 0.5)).half()
for _ in range(8):
    DN.conv3x3_hip(x, w)
torch.cuda.synchronize()
