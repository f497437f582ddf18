"""The CPU oracle (oracle/ref_cpu.py) against the goldens produced by the reference's own code
(tests/golden/make_golden.py).  Runs anywhere torch-CPU is installed."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from video_dqn_amd import synth
from helpers import g4_inputs


def test_g1_state_dict_layout(g1):
    for ec in (True, False):
        for pano in (True, False):
            ref = g1[f"ec{int(ec)}_pano{int(pano)}"]
            m = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=ec, panorama=pano)
            sd = m.state_dict()
            assert list(sd.keys()) == ref["keys"]
            assert [list(v.shape) for v in sd.values()] == ref["shapes"]
            assert [n for n, _ in m.named_parameters()] == ref["params"]
            assert len(ref["keys"]) == (250 if ec else 244)
            # synth.make_state_dict produces the same key set and loads strictly
            F = 4 if pano else 1
            s = synth.make_state_dict(3, extra_capacity=ec, num_frames=F)
            assert list(s.keys()) == ref["keys"]
            m.load_state_dict(s, strict=True)


def test_g2_forward_matches_reference(golden):
    for ec, pano, B, st in golden["g2_cases"]:
        ec, pano, st = bool(ec), bool(pano), bool(st)
        F = 4 if pano else 1
        m = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=ec, panorama=pano)
        m.load_state_dict(synth.make_state_dict(11, extra_capacity=ec, num_frames=F))
        (tup, _) = synth.make_batch(21 + int(B), int(B), F, structured=True)
        mode = "set_train" if st else "eval"
        m.set_train() if st else m.eval()
        with torch.no_grad():
            q = m(tup[0]).numpy()
        ref = golden[f"g2_q_ec{int(ec)}_pano{int(pano)}_B{int(B)}_{mode}"]
        assert q.shape == ref.shape == (int(B), 5, 3)
        np.testing.assert_allclose(q, ref, rtol=1e-5, atol=1e-6)
        bn = [int(mod.training) for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        assert bn == list(golden[f"g2_bntrain_ec{int(ec)}_pano{int(pano)}_B{int(B)}_{mode}"])


def test_g3_td_steps_match_reference(golden):
    torch.set_num_threads(8)
    cfg = ref_cpu.default_config()
    tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7))
    tr.target_net.load_state_dict(synth.make_state_dict(8))
    names = [n for n, _ in tr.model.named_parameters()]
    assert names == list(golden["g3_param_names"])
    for step in (1, 2, 3):
        (tup, _) = synth.make_batch(100 + step, 8, 1, structured=True, reward_p=0.3)
        d = {}
        loss = tr.step(tup, d)
        np.testing.assert_allclose(loss, float(golden[f"g3_loss_s{step}"]), rtol=1e-5)
        np.testing.assert_allclose(d["before_values"].detach().numpy(), golden[f"g3_qbefore_s{step}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(d["after_values"].numpy(), golden[f"g3_qafter_target_s{step}"], rtol=1e-4, atol=1e-6)
        for n, p in tr.model.named_parameters():
            if p.grad is None:
                assert f"g3_gradnone_{n}" in golden.files
                continue
            g = p.grad.flatten()
            idx = synth.randint(1234, "idx." + n, (min(16, g.numel()),), g.numel())
            amax = float(golden[f"g3_gabsmax_s{step}_{n}"])
            np.testing.assert_allclose(g[idx].numpy(), golden[f"g3_gsamp_s{step}_{n}"], rtol=1e-3, atol=1e-5 * amax)
            np.testing.assert_allclose(g.double().norm().item(), float(golden[f"g3_gnorm_s{step}_{n}"]), rtol=1e-4)
    assert int(golden["g3_bn_unchanged"]) == 1
    ost = tr.optimizer.state_dict()
    assert list(golden["g3_opt_param_ids"]) == ost["param_groups"][0]["params"]
    assert list(golden["g3_opt_state_ids"]) == sorted(ost["state"].keys())
    assert "g3_gradnone_resnet.fc.weight" in golden.files and "g3_gradnone_resnet.fc.bias" in golden.files


def test_g4_loss_branches_match_reference(golden):
    for cid, clip, linear, rbr, gamma, s in golden["g4_cases"]:
        cfg = ref_cpu.default_config(LOSS_CLIP=("none", "rect", "sigmoid")[int(clip)], LINEAR=bool(linear),
                                     REMOVE_BEFORE_REWARD=bool(rbr), GAMMA=float(gamma))
        qb, qo, qt, act, rew, term, vm = g4_inputs(int(s))
        loss, dq, best, y = ref_cpu.td_loss_from_q(qb, qo, qt, act, rew, term, vm, cfg)
        np.testing.assert_allclose(loss.item(), float(golden[f"g4_loss_{int(cid)}"]), rtol=1e-6)
        np.testing.assert_allclose(dq.numpy(), golden[f"g4_dq_{int(cid)}"], rtol=1e-6, atol=1e-8)
        assert best[0, 0].item() == 0 and best[1, 1].item() == 1  # first max wins


BASIC_CASES = (("F1", False, 1, 6, 2), ("F4", True, 4, 3, 1))  # tests/golden/make_golden_basic.py


def test_g5_basic_arch_td_steps_match_reference(golden_basic):
    """ARCHITECTURE='basic' (train-mode BatchNorm, per-frame-slot batch statistics, running-stat updates): the oracle
    restatement against goldens produced by the reference class + the reference's process_batch."""
    torch.set_num_threads(8)
    g = golden_basic
    for tag, pano, F, B, steps in BASIC_CASES:
        cfg = ref_cpu.default_config(ARCHITECTURE="basic", PANORAMA=pano)
        tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7, extra_capacity=False, num_frames=F))
        tr.target_net.load_state_dict(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
        for step in range(1, steps + 1):
            (tup, _) = synth.make_batch(400 + 10 * F + step, B, F, structured=True, reward_p=0.3)
            d = {}
            loss = tr.step(tup, d)
            k = f"g5_{tag}_s{step}"
            np.testing.assert_allclose(loss, float(g[f"{k}_loss"]), rtol=1e-5)
            np.testing.assert_allclose(d["before_values"].detach().numpy(), g[f"{k}_qbefore"], rtol=1e-4, atol=1e-6)
            for n, v in tr.model.state_dict().items():
                if n.startswith("features."):
                    continue
                if "running_" in n:
                    np.testing.assert_allclose(v.numpy(), g[f"{k}_bn_{n}"], rtol=1e-5, atol=1e-7, err_msg=n)
                elif "num_batches_tracked" in n:
                    assert int(v) == int(g[f"{k}_nbt_{n}"]) == 2 * F * step
            for n, p in tr.model.named_parameters():
                if p.grad is None:
                    continue
                np.testing.assert_allclose(p.grad.double().norm().item(), float(g[f"{k}_gnorm_{n}"]), rtol=1e-4)
        tr.model.eval()
        with torch.no_grad():
            (tup, _) = synth.make_batch(499, 2, F, structured=True)
            np.testing.assert_allclose(tr.model(tup[0]).numpy(), g[f"g5_{tag}_eval_q_after_training"], rtol=1e-4, atol=1e-6)
