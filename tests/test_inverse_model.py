"""Inverse-action model (archs/inverse_action2.py:45-100): goldens G8/G9 come from the reference
class itself (tests/golden/make_golden_inverse.py).  CPU: the oracle restatement and the product's state_dict layout;
GPU: the product's eval forward through the C ABI."""
import os

import numpy as np
import pytest
import torch

from video_dqn_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ginv():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_inverse.npz"), allow_pickle=False)


def _frames(B):
    be = synth.normalise_frames(synth.make_frames_uint8(50 + B, "be", B, 1, structured=True))
    ae = synth.normalise_frames(synth.make_frames_uint8(50 + B, "ae", B, 1, structured=True))
    return be, ae


def test_oracle_and_layout_match_reference(ginv):
    from oracle import ref_cpu
    from video_dqn_amd.inverse_model import InverseActionModel
    m = ref_cpu.InverseActionModel()
    sd = m.state_dict()
    assert list(sd.keys()) == list(ginv["g8_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(ginv["g8_shapes"])
    w = synth.make_inverse_state_dict(21)
    assert list(w.keys()) == list(ginv["g8_keys"])
    m.load_state_dict(w, strict=True)
    m.eval()
    for B in (1, 4):
        with torch.no_grad():
            enc, y = m(*_frames(B))
        np.testing.assert_allclose(enc.numpy(), ginv[f"g9_enc_B{B}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(y.numpy(), ginv[f"g9_y_B{B}"], rtol=1e-4, atol=1e-6)
    # the product model (storage-only on CPU) exposes the same 132 keys / shapes and loads the checkpoint strictly
    p = InverseActionModel(device="cpu")
    psd = p.state_dict()
    assert list(psd.keys()) == list(ginv["g8_keys"])
    assert [str(tuple(v.shape)) for v in psd.values()] == list(ginv["g8_shapes"])
    p.load_state_dict(w, strict=True)
    for k, v in p.state_dict().items():
        assert torch.equal(v.cpu(), w[k]), k
    with pytest.raises(RuntimeError):
        p.load_state_dict({k: v for k, v in w.items() if k != "fc2.bias"}, strict=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("bf16", 5e-2)])
def test_gpu_forward_matches_reference_golden(ginv, dtype, tol):
    from video_dqn_amd.inverse_model import InverseActionModel
    m = InverseActionModel(dtype=dtype, device="cuda", max_batch=4)
    m.load_state_dict(synth.make_inverse_state_dict(21), strict=True)
    m.eval()
    for B in (1, 4):
        be, ae = _frames(B)
        enc, y = m(be.cuda(), ae.cuda())
        torch.cuda.synchronize()
        ref_y, ref_enc = torch.from_numpy(ginv[f"g9_y_B{B}"]), torch.from_numpy(ginv[f"g9_enc_B{B}"])
        assert ((y.cpu() - ref_y).abs().max() / ref_y.abs().max()).item() < tol
        assert (enc.cpu() - ref_enc).abs().max().item() < tol
        if dtype == "f32":
            assert list(y.argmax(dim=1).cpu().numpy()) == list(ginv[f"g9_act_B{B}"])  # the labels the data pipeline stores
    # uint8 frames (normalisation fused into the input kernel) give the same answer as the normalised tensors
    fb = synth.make_frames_uint8(54, "be", 4, 1, structured=True)[:, 0]
    fa = synth.make_frames_uint8(54, "ae", 4, 1, structured=True)[:, 0]
    enc8, y8 = m(torch.from_numpy(fb).cuda(), torch.from_numpy(fa).cuda())
    assert ((y8.cpu() - torch.from_numpy(ginv["g9_y_B4"])).abs().max() / torch.from_numpy(ginv["g9_y_B4"]).abs().max()).item() < tol
    m.train()
    with pytest.raises(Exception):
        m(be.cuda(), ae.cuda())


@pytest.mark.gpu
def test_gpu_training_steps_match_reference_golden():
    """Two optimisation steps of train_inverse_model.py (reference model class + loop statements, goldens G10) with the
    dropout masks of the golden run replayed: loss, logits, every head gradient, post-Adam parameters (f32)."""
    from video_dqn_amd.inverse_model import InverseActionModel
    from video_dqn_amd.inverse_train import InverseTrainer
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_inverse_train.npz"), allow_pickle=False)
    m = InverseActionModel(dtype="f32", device="cuda", max_batch=8)
    m.load_state_dict(synth.make_inverse_state_dict(21), strict=True)
    tr = InverseTrainer(m, lr=1e-4, weight_decay=0.0)
    assert tr.names == list(g["g10_trainable"])
    B, lr = 6, 1e-4
    for step in (1, 2):
        be = synth.normalise_frames(synth.make_frames_uint8(70 + step, "be", B, 1, structured=True))
        ae = synth.normalise_frames(synth.make_frames_uint8(70 + step, "ae", B, 1, structured=True))
        act = torch.from_numpy(synth.randint(70 + step, "act", (B,), 3))
        loss, y = tr.step(be.cuda(), ae.cuda(), act.cuda(), torch.from_numpy(g[f"g10_s{step}_mask"]))
        torch.cuda.synchronize()
        k = f"g10_s{step}"
        np.testing.assert_allclose(loss.item(), float(g[f"{k}_loss"]), rtol=1e-3)
        ref_y = torch.from_numpy(g[f"{k}_y"])
        assert ((y.cpu() - ref_y).abs().max() / ref_y.abs().max()).item() < 1e-3
        for n in tr.names:
            gr = tr.gviews[n].flatten().cpu()
            idx = synth.randint(1234, "idx." + n, (min(16, gr.numel()),), gr.numel())
            amax = float(g[f"{k}_gabsmax_{n}"])
            if step == 1:
                np.testing.assert_allclose(gr.double().norm().item(), float(g[f"{k}_gnorm_{n}"]), rtol=2e-3, err_msg=n)
                assert np.abs(gr[idx].numpy() - g[f"{k}_gsamp_{n}"]).max() <= 3e-3 * amax + 1e-12, n
            pd = np.abs(tr.views[n].flatten().cpu()[idx].numpy() - g[f"{k}_psamp_{n}"])
            assert pd.max() <= 2.5 * lr * step, n  # Adam's first steps are sign-like: at most one flipped sign per step
    # the trained head is what inference now uses
    m.eval()
    enc, y = m(be.cuda(), ae.cuda())
    assert torch.isfinite(y).all() and abs(enc.sum().item() - B) < 1e-3
