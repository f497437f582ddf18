#!/usr/bin/env python
"""Numerics of Winograd F(2x2, 3x3) with bf16 matrix operands against the direct convolution the engine computes (VERDICT round 5, item 8,
second number).  CPU, torch only: the MFMA would multiply bf16 TRANSFORMED inputs (B^T d B, rounded to bf16) with bf16 TRANSFORMED weights
(G g G^T, rounded to bf16) and accumulate in f32; the output transform A^T m A is f32.  Reported: max error / max |reference| of the f32
output against F.conv2d on the same bf16 operands in float64 — the figure tests/test_gpu_ops.py holds the engine's kernels to at 2e-4
(f32 output of identical bf16 operands: only the summation order differs there)."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def bf16(t):
    return t.float().to(torch.bfloat16).double()


def winograd(x, w, round_ops):
    n, c, h, wd = x.shape
    k = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = h // 2, wd // 2
    # tiles: [n, c, th, tw, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)
    if round_ops:
        V, U = bf16(V), bf16(U)
    M = torch.einsum("nctuil,kcil->nktuil", V.float().double(), U.float().double()) if not round_ops else torch.einsum("nctuil,kcil->nktuil", V, U)
    Y = torch.einsum("ij,nktujl,ml->nktuim", AT, M, AT)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)


def main():
    for name, c, hw, n in (("layer2 128->128 @28", 128, 28, 4), ("layer3 256->256 @14", 256, 14, 8), ("layer4 512->512 @7 (8x8 padded)", 512, 8, 8)):
        x = bf16(torch.relu(torch.randn(n, c, hw, hw)))          # post-ReLU activations, already bf16 in the engine
        w = bf16(torch.randn(c, c, 3, 3) * (2.0 / (9 * c)) ** 0.5)  # He-scaled weights, bf16 packed operands
        ref = F.conv2d(x, w, padding=1)
        exact = winograd(x, w, False)
        wino = winograd(x, w, True)
        f32_order = F.conv2d(x.float(), w.float(), padding=1).double()  # what a different f32 summation order costs (the present gate's content)
        s = ref.abs().max()
        print(f"{name}: transform identity {(exact - ref).abs().max() / s:.1e} | f32 accumulation of the direct form {(f32_order - ref).abs().max() / s:.1e} | "
              f"Winograd with bf16 transformed operands: max {(wino - ref).abs().max() / s:.2e}, relative L2 {((wino - ref).norm() / ref.norm()):.2e} "
              f"(one bf16 rounding of the output alone: max {(bf16(ref) - ref).abs().max() / s:.2e}, L2 {((bf16(ref) - ref).norm() / ref.norm()):.2e})")


if __name__ == "__main__":
    main()
