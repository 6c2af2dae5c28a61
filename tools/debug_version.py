import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
for fused in (True, False):
    for foreach in (None,):
        p = torch.nn.Parameter(torch.randn(1000, device=dev))
        opt = torch.optim.AdamW([p], lr=1e-2, fused=fused)
        p.grad = torch.randn_like(p)
        v0 = p._version
        opt.step()
        print("AdamW fused" if fused else "AdamW foreach", "version", v0, "->", p._version)
