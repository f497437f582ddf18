#!/usr/bin/env python
"""Localise the run-to-run spread of the ARCHITECTURE='basic' f32 update in the default (atomic-sum) mode (one draw
in seven had eng_max 2.73e-2 against 1.59e-2 .. 1.84e-2 for the rest — summation-order chaos of train-mode BatchNorm, or an
ordering bug of the two-stream schedule?).  N runs of the same update in each of three modes:
    atomic + overlap (the default), atomic + VDQN_NO_OVERLAP (one stream: no cross-stream ordering left), deterministic
Per run: the worst gradient element vs the float64 oracle (tensor, flat index, error / tensor max), the worst tensor by L2, and
the run's distance to the DETERMINISTIC run (if the spread is rounding chaos it shows in both overlap settings and the distance to
the deterministic result has the same distribution; a race shows only with the overlap on, and as O(1) element differences)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu  # noqa: E402
from video_dqn_amd import synth  # noqa: E402
from video_dqn_amd.engine import NetEngine, TDStepper  # noqa: E402


def main(F=4, B=3, runs=30):
    dev = "cuda"
    (tup, _) = synth.make_batch(400 + 10 * F + 1, B, F, structured=True, reward_p=0.3)
    cfg = ref_cpu.default_config(ARCHITECTURE="basic", PANORAMA=F > 1)
    ref = {}
    for prec in (torch.float32, torch.float64):
        tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7, extra_capacity=False, num_frames=F), num_frames=F)
        tr.target_net.load_state_dict(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
        tr.model.to(prec), tr.target_net.to(prec)
        tr.model.set_train()
        ref_cpu.process_batch(tr.model, tr.target_net, cfg, (tup[0].to(prec), tup[1].to(prec)) + tuple(tup[2:])).backward()
        ref[prec] = {n: p.grad.double() for n, p in tr.model.named_parameters() if p.grad is not None}
    g64 = ref[torch.float64]
    own = max(((ref[torch.float32][n] - r).abs().max() / r.abs().max()).item() for n, r in g64.items())
    print(f"F={F} B={B}: fp32 oracle's own worst element vs float64: {own:.4e}; 1.5x line {1.5 * own:.4e}, 2x line {2 * own:.4e}")
    args = (tup[0].contiguous().to(dev), tup[1].contiguous().to(dev), 1, tup[2].to(dev), tup[3].float().to(dev), tup[4].float().to(dev))

    def one(det, overlap):
        net = NetEngine(3, 5, F, False, "f32", 2 * B, deterministic=det)
        net.load_tensors(synth.make_state_dict(7, extra_capacity=False, num_frames=F))
        stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
        tnet = NetEngine(3, 5, F, False, "f32", 2 * B)
        tnet.load_tensors(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
        tnet.pack_weights(stp.packed_target)
        net.lib.vdqn_net_set_overlap(net.handle, overlap)
        stp.forward_backward(*args)
        torch.cuda.synchronize()
        return net, stp.grads.cpu().double()

    net, g_det = one(True, 1)

    def describe(g):
        worst_el, worst_l2 = (0.0, "", -1), (0.0, "")
        for n, r in g64.items():
            s = net.slots[n]
            ge = g[s.offset:s.offset + s.numel].view(s.shape)
            d = (ge - r).abs()
            e = (d.max() / r.abs().max()).item()
            if e > worst_el[0]:
                worst_el = (e, n, int(d.argmax()))
            l2 = ((ge - r).norm() / r.norm()).item()
            if l2 > worst_l2[0]:
                worst_l2 = (l2, n)
        return worst_el, worst_l2

    we, wl = describe(g_det)
    print(f"deterministic: worst element {we[0]:.4e} ({we[1]}[{we[2]}]), worst L2 {wl[0]:.4e} ({wl[1]})")
    for label, overlap in (("atomic+overlap", 1), ("atomic+no_overlap", 0)):
        emax = []
        for i in range(runs):
            _, g = one(False, overlap)
            we, wl = describe(g)
            dd = ((g - g_det).norm() / g_det.norm()).item()
            dmax = ((g - g_det).abs().max() / g_det.abs().max()).item()
            emax.append(we[0])
            print(f"{label} run {i:2d}: worst element {we[0]:.4e} ({we[1]}[{we[2]}]) worst L2 {wl[0]:.4e} ({wl[1]}) | vs deterministic run: L2 {dd:.3e} max {dmax:.3e}")
        t = torch.tensor(emax)
        print(f"{label}: worst-element error over {runs} runs: min {t.min():.4e} median {t.median():.4e} max {t.max():.4e}; above 1.5x line: "
              f"{int((t > 1.5 * own).sum())}, above 2x line: {int((t > 2 * own).sum())}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 30)
