"""What the extra tensors of the implicit-GEMM epilogues cost per launch (isolated launches, HIP events of the launch profiler): the data
gradient with the ReLU mask, the column sums and the residual gradient; the forward pass with the residual."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from video_dqn_amd import _lib, ops
dev="cuda"; dt=torch.bfloat16
def run(name, n, hw, ch, **extra):
    x = torch.randn((n, hw, hw, ch), device=dev).to(dt)
    w = (torch.randn((ch, 3, 3, ch), device=dev) * 0.05).to(dt)
    kw = dict(ho=hw, wo=hw, co=ch, r=3, s=3, stride=1, pad=1, mode=1)
    if extra.get("mask"): kw["mask"] = torch.randn((n, hw, hw, ch), device=dev).to(dt)
    if extra.get("resid"): kw["resid"] = torch.randn((n, hw, hw, ch), device=dev).to(dt)
    if extra.get("colsum"): kw["want_colsum"] = True
    for _ in range(3): ops.conv2d(x, w, **kw)
    torch.cuda.synchronize(); _lib.profile_enable(True)
    for _ in range(20): ops.conv2d(x, w, **kw)
    torch.cuda.synchronize(); prof = _lib.profile_collect(); _lib.profile_enable(False)
    for tag, v in prof.items():
        print(f"{name:28s} {str(extra):40s} {tag:28s} {1e3*v['ms']/v['launches']:7.1f} us")
def run_fwd(name, n, hw, ch, resid):
    x = torch.randn((n, hw, hw, ch), device=dev).to(dt)
    w = (torch.randn((ch, 3, 3, ch), device=dev) * 0.05).to(dt)
    kw = dict(ho=hw, wo=hw, co=ch, r=3, s=3, stride=1, pad=1, relu=True)
    if resid: kw["resid"] = torch.randn((n, hw, hw, ch), device=dev).to(dt)
    for _ in range(3): ops.conv2d(x, w, **kw)
    torch.cuda.synchronize(); _lib.profile_enable(True)
    for _ in range(20): ops.conv2d(x, w, **kw)
    torch.cuda.synchronize(); prof = _lib.profile_collect(); _lib.profile_enable(False)
    for tag, v in prof.items():
        print(f"{name:28s} {'resid' if resid else 'plain':40s} {tag:28s} {1e3*v['ms']/v['launches']:7.1f} us")
for name, n, hw, ch in (("layer2 fwd 128@28", 512, 28, 128), ("layer2 fwd 128@28", 256, 28, 128), ("layer3 fwd 256@14", 512, 14, 256)):
    for r in (0, 1):
        run_fwd(name + f" n{n}", n, hw, ch, r)
for name, n, hw, ch in (("layer2 dgrad 128@28", 256, 28, 128), ("layer3 dgrad 256@14", 256, 14, 256), ("layer1 dgrad 64@56", 256, 56, 64)):
    for extra in ({}, {"mask": 1}, {"mask": 1, "colsum": 1}, {"mask": 1, "colsum": 1, "resid": 1}):
        run(name, n, hw, ch, **extra)
