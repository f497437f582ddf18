"""The Q-head's skinny GEMM kernels (csrc/skinny.hip; bf16): `top` Linear layers forward / data gradient and the features.8
valid 3x3 convolution (archs/HabitatDQNMultiAction.py:30-31) through vdqn_conv2d, against torch on the same bf16 operands
(f32 accumulation both sides: 2e-4 of the tensor's max for f32 outputs, one bf16 ulp class — 1e-2 — for bf16 outputs), and
against the generic tiled kernel (VDQN_SKINNY=0 in a child process is what tests/test_gpu_engine.py's switch matrix covers)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import relerr  # noqa: E402
from video_dqn_amd import synth  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rnd(seed, name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(seed, name, shape, lo, hi))


@pytest.mark.parametrize("B,fin,fout", [(512, 1600, 512), (256, 512, 256), (37, 256, 15), (1, 6400, 512), (300, 19200, 512)])
def test_linear_forward(B, fin, fout):
    from video_dqn_amd import _lib, ops
    x = rnd(1, "x", (B, fin)).to(BF)
    w = rnd(2, "w", (fout, fin), -0.05, 0.05).to(BF)
    b = rnd(3, "b", (fout,))
    cop = (fout + 63) // 64 * 64
    wp = torch.zeros((cop, fin), dtype=BF); wp[:fout] = w
    bp = torch.zeros(cop); bp[:fout] = b
    out, outf = ops.conv2d(x.view(B, 1, 1, fin).to(DEV), wp.view(cop, 1, 1, fin).to(DEV), ho=1, wo=1, co=cop, r=1, s=1, stride=1, pad=0,
                           bias=bp.to(DEV), relu=True, want_f32=True)
    torch.cuda.synchronize()
    ref = torch.relu(F.linear(x.float(), w.float(), b))
    assert relerr(outf.view(B, cop)[:, :fout], ref) < 2e-4
    assert relerr(out.view(B, cop)[:, :fout], ref) < 1e-2
    assert float(outf.view(B, cop)[:, fout:].abs().max()) == 0.0 if cop > fout else True
    assert torch.equal(out.view(B, cop).float(), outf.view(B, cop).to(BF).float())  # the bf16 store is the rounding of the f32 value


@pytest.mark.parametrize("B,fin,fout", [(256, 512, 1600), (256, 256, 512), (100, 64, 256), (16, 512, 19200)])
def test_linear_dgrad_with_mask_and_column_sums(B, fin, fout):
    """gx = (gy @ Wd^T) masked by the saved activation, plus the per-tile column sums of the stored values (the bias gradient of
    the layer below): vdqn_conv2d_colsum_rows tells the buffer's row granularity."""
    from video_dqn_amd import _lib, ops
    import ctypes as C
    gy = rnd(5, "gy", (B, fin)).to(BF)
    wd = rnd(6, "wd", (fout, fin), -0.05, 0.05).to(BF)   # data-gradient operand: [in_features of the layer][out_features]
    act = rnd(7, "act", (B, fout)).clamp_min(0).to(BF)    # post-ReLU activation of the layer below (zeros = masked)
    gx, part = ops.conv2d(gy.view(B, 1, 1, fin).to(DEV), wd.view(fout, 1, 1, fin).to(DEV), ho=1, wo=1, co=fout, r=1, s=1, stride=1, pad=0,
                          mode=1, mask=act.view(B, 1, 1, fout).to(DEV), want_colsum=True)
    torch.cuda.synchronize()
    ref = (gy.float() @ wd.float().t()) * (act.float() > 0)
    assert relerr(gx.view(B, fout), ref) < 1e-2
    assert part.shape[0] == (B + 31) // 32  # skinny kernel: one entry per 32 rows
    stored = gx.view(B, fout).float()
    assert relerr(part.sum(0), stored.sum(0).to(DEV)) < 1e-5
    assert float(gx.view(B, fout).float()[act.to(DEV).float() <= 0].abs().max()) == 0.0


@pytest.mark.parametrize("n", [512, 3, 77, 1])
def test_features8_valid_conv_forward(n):
    from video_dqn_amd import ops
    x = rnd(11, "x", (n, 7, 7, 512)).to(BF)
    w = rnd(12, "w", (64, 512, 3, 3), -0.02, 0.02).to(BF)
    b = rnd(13, "b", (64,))
    wp = w.permute(0, 2, 3, 1).contiguous()  # [co][r][s][ci]
    out = ops.conv2d(x.to(DEV), wp.to(DEV), ho=5, wo=5, co=64, r=3, s=3, stride=1, pad=0, bias=b.to(DEV), relu=True)
    torch.cuda.synchronize()
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b)).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-2
    assert relerr(out.float(), ref.to(BF).float()) < 8e-3  # at most one bf16 ulp off the rounded reference


@pytest.mark.parametrize("n,hi,ci", [(5, 56, 64), (3, 28, 128), (7, 14, 256), (1, 2, 64), (64, 56, 64)])
def test_downsample_1x1_stride2(n, hi, ci):
    """The BasicBlock downsample convolution (1x1 / stride 2, C -> 2C, BatchNorm folded into weights + bias, no ReLU) against torch
    on the same bf16 operands (tiled kernel; the streaming variant of round 4 measured slower and lives in experiments/)."""
    from video_dqn_amd import ops
    co, ho = 2 * ci, hi // 2
    x = rnd(21, "x", (n, hi, hi, ci)).to(BF)
    w = rnd(22, "w", (co, ci), -0.1, 0.1).to(BF)
    b = rnd(23, "b", (co,))
    out = ops.conv2d(x.to(DEV), w.view(co, 1, 1, ci).to(DEV), ho=ho, wo=ho, co=co, r=1, s=1, stride=2, pad=0, bias=b.to(DEV))
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, ci, 1, 1), b, stride=2).permute(0, 2, 3, 1)
    assert out.shape == ref.shape
    assert relerr(out, ref) < 1e-2
    assert relerr(out.float(), ref.to(BF).float()) < 8e-3


def test_skinny_switch_off_gives_the_tiled_kernels_granularity():
    """VDQN_SKINNY=0 (read once per process) routes the same calls to the generic 128-row tiles."""
    import os
    import subprocess
    import sys
    code = ("import torch, ctypes as C\n"
            "from video_dqn_amd import _lib\n"
            "a = _lib.ConvArgs(); a.n_img = 256; a.hi = a.wi = a.ho = a.wo = 1; a.ci = a.pix_stride = 512; a.co = a.ldo = 256\n"
            "a.r = a.s = a.stride = 1; a.dtype = _lib.VDQN_BF16; a.mode = 1\n"
            "print(_lib.load().vdqn_conv2d_colsum_rows(C.byref(a)))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env, want in (({}, "32"), ({"VDQN_SKINNY": "0"}, "128")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == want, r.stdout
