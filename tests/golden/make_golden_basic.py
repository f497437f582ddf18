#!/usr/bin/env python
"""Golden vectors for ARCHITECTURE='basic' (defaults.py:14), produced by the REFERENCE's own class and its own
``process_batch`` statements (same import machinery as make_golden.py; build container only).

'basic' keeps the whole ResNet in train mode (archs/HabitatDQNMultiAction.py:37-40 only puts it in eval mode for
extra_capacity), so every online forward normalises with batch statistics — per frame slot, because ``features`` is
applied slot by slot (:49-51) — and updates the running statistics.  G5 pins, for one and two TD updates:
loss, Q(s), gradient norms/samples of every parameter, post-Adam parameter samples, running statistics and
``num_batches_tracked`` after the update.

Usage:  python tests/golden/make_golden_basic.py      (writes tests/golden/golden_basic.npz)
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import extract_process_batch, import_reference_model, sample_idx  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from video_dqn_amd import synth  # noqa: E402

CASES = (("F1", False, 1, 6, 2), ("F4", True, 4, 3, 1))  # tag, panorama, F, B, steps


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefModel = import_reference_model()
    pb_factory = extract_process_batch()
    out = {}
    for tag, pano, F, B, steps in CASES:
        cfg = ref_cpu.default_config(ARCHITECTURE="basic", PANORAMA=pano)
        cfg.device = torch.device("cpu")
        sd = synth.make_state_dict(7, extra_capacity=False, num_frames=F)
        model = RefModel(3, 5, extra_capacity=False, panorama=pano)
        model.load_state_dict(sd)
        target = RefModel(3, 5, extra_capacity=False, panorama=pano)
        target.load_state_dict(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
        target.eval()  # train_q_network.py:122
        opt = torch.optim.Adam(model.parameters(), lr=cfg.LEARNING_RATE)  # :124
        process_batch = pb_factory(model, target, cfg)
        first = {}

        def keep_first(mod, inp, outp):  # Q(s) is the first model call of process_batch (:131); must return None
            if "q" not in first:
                first["q"] = outp.detach().clone()
        model.register_forward_hook(keep_first)
        for step in range(1, steps + 1):
            (tup, _) = synth.make_batch(400 + 10 * F + step, B, F, structured=True, reward_p=0.3)
            first.clear()
            model.set_train()  # :221
            opt.zero_grad()  # :222
            loss = process_batch(tup, compare_ground_truth=False, batch_number=step)  # :223
            loss.backward()  # :226
            k = f"g5_{tag}_s{step}"
            out[f"{k}_loss"] = np.array(loss.item(), dtype=np.float64)
            out[f"{k}_qbefore"] = first["q"].numpy()
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                g = p.grad.detach().flatten()
                idx = sample_idx(n, g.numel())
                out[f"{k}_gnorm_{n}"] = np.array(g.double().norm().item())
                out[f"{k}_gabsmax_{n}"] = np.array(g.abs().max().item())
                out[f"{k}_gsamp_{n}"] = g[idx].numpy()
            if step == 1:
                # the same update by the same reference code in float64: how far the reference's OWN fp32 run is from
                # exact arithmetic (train-mode BatchNorm makes a ReLU sign flip non-local) — the yardstick of the GPU gate
                m64 = RefModel(3, 5, extra_capacity=False, panorama=pano)
                m64.load_state_dict(sd)
                t64 = RefModel(3, 5, extra_capacity=False, panorama=pano)
                t64.load_state_dict(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
                m64.double()
                t64.double().eval()
                m64.set_train()
                tup64 = (tup[0].double(), tup[1].double()) + tuple(tup[2:])
                pb_factory(m64, t64, cfg)(tup64, compare_ground_truth=False, batch_number=1).backward()
                e_max, e_l2 = 0.0, 0.0
                g32 = dict(model.named_parameters())
                for n, p in m64.named_parameters():
                    if p.grad is None:
                        continue
                    g = p.grad.detach().flatten()
                    out[f"{k}_gsamp64_{n}"] = g[sample_idx(n, g.numel())].numpy()
                    out[f"{k}_gnorm64_{n}"] = np.array(g.norm().item())
                    out[f"{k}_gabsmax64_{n}"] = np.array(g.abs().max().item())
                    d = g32[n].grad.detach().flatten().double() - g
                    e_max = max(e_max, (d.abs().max() / g.abs().max()).item())
                    e_l2 = max(e_l2, (d.norm() / g.norm()).item())
                out[f"{k}_ref32_vs_ref64_worst_max"] = np.array(e_max)
                out[f"{k}_ref32_vs_ref64_worst_l2"] = np.array(e_l2)
                print(tag, "reference fp32 vs fp64: worst max-err", e_max, "worst l2-err", e_l2)
            opt.step()  # :227
            for n, p in model.named_parameters():
                out[f"{k}_psamp_{n}"] = p.detach().flatten()[sample_idx(n, p.numel())].numpy()
            for n, v in model.state_dict().items():
                if n.startswith("features."):
                    continue  # aliases of resnet.*
                if "running_" in n:
                    out[f"{k}_bn_{n}"] = v.numpy().copy()
                elif "num_batches_tracked" in n:
                    out[f"{k}_nbt_{n}"] = np.array(int(v))
        model.eval()
        with torch.no_grad():
            (tup, _) = synth.make_batch(499, 2, F, structured=True)
            out[f"g5_{tag}_eval_q_after_training"] = model(tup[0]).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_basic.npz"), **out)
    print("wrote golden_basic.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
