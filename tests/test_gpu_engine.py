"""End-to-end parity of the HIP engine (through the C ABI) with the CPU oracle and with the committed goldens
(which were produced by the reference's own code): Q-values, loss, gradients, post-Adam parameters.

Tolerance: north_star's 1e-3 relative (fp32) — applied per tensor relative to the tensor's max |value|.
The bf16 throughput mode is compared with the same fp32 oracle at a looser, stated tolerance."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import relerr  # noqa: E402
from video_dqn_amd import synth  # noqa: E402

DEV = "cuda"


def make_engine(dtype, seed=7, num_frames=1, max_batch=32, deterministic=None):
    from video_dqn_amd.engine import NetEngine
    net = NetEngine(3, 5, num_frames, True, dtype, max_batch, deterministic=deterministic)
    net.load_tensors(synth.make_state_dict(seed, num_frames=num_frames))
    return net


def test_param_table_matches_reference_layout(g1):
    net = make_engine("f32")
    ref = g1["ec1_pano0"]
    trainable = [s for s in net.slots.values() if s.kind in (0, 1)]
    by_id = sorted(trainable, key=lambda s: s.param_id)
    assert [s.name for s in by_id] == ref["params"]            # model.parameters() order == Adam ids
    shapes = dict(zip(ref["keys"], ref["shapes"]))
    for s in net.slots.values():
        assert list(s.shape) == shapes[s.name]
    assert net.trainable_numel >= 12426383 and net.params_numel - net.trainable_numel == 513000


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("bf16", 4e-2)])
@pytest.mark.parametrize("num_frames,B", [(1, 3), (4, 2)])
def test_forward_matches_oracle(dtype, tol, num_frames, B):
    from oracle import ref_cpu
    net = make_engine(dtype, seed=11, num_frames=num_frames)
    (tup, raw) = synth.make_batch(21 + B, B, num_frames, structured=True)
    m = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=num_frames > 1, num_frames=num_frames)
    m.load_state_dict(synth.make_state_dict(11, num_frames=num_frames))
    m.eval()
    with torch.no_grad():
        ref = m(tup[0]).reshape(B, 15)
    q1 = net.forward(tup[0].contiguous().to(DEV), 1, B)           # f32 NCHW normalised frames
    q0 = net.forward(torch.from_numpy(raw[0]).to(DEV), 0, B)       # uint8 NHWC frames, normalise fused
    torch.cuda.synchronize()
    assert relerr(q1, ref) < tol
    assert relerr(q0, ref) < tol


def test_forward_matches_reference_golden(golden):
    """G2 goldens came from the reference class itself (tests/golden/make_golden.py)."""
    for ec, pano, B, st in golden["g2_cases"]:
        if not ec:
            continue
        F = 4 if pano else 1
        net = make_engine("f32", seed=11, num_frames=F)
        (tup, _) = synth.make_batch(21 + int(B), int(B), F, structured=True)
        q = net.forward(tup[0].contiguous().to(DEV), 1, int(B))
        torch.cuda.synchronize()
        ref = torch.from_numpy(golden[f"g2_q_ec1_pano{int(pano)}_B{int(B)}_{'set_train' if st else 'eval'}"]).reshape(int(B), 15)
        assert relerr(q, ref) < 1e-3


def _run_steps(dtype, steps, B=8, batch_seed0=100):
    from video_dqn_amd.engine import TDStepper
    net = make_engine(dtype, seed=7, max_batch=2 * B)
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    # target network = a different parameter set, as in the golden run
    tnet = make_engine(dtype, seed=8, max_batch=2 * B)
    tnet.pack_weights(stp.packed_target)
    out = []
    for step in range(1, steps + 1):
        (tup, raw) = synth.make_batch(batch_seed0 + step, B, 1, structured=True, reward_p=0.3)
        before, after, act, rew, term, gt, vm = tup
        loss = stp.step(before.contiguous().to(DEV), after.contiguous().to(DEV), 1, act.to(DEV), rew.float().to(DEV),
                        term.float().to(DEV))
        torch.cuda.synchronize()
        out.append(dict(loss=loss.item(), q_before=stp.q_before.cpu().clone(), grads=stp.grads.cpu().clone(),
                        params=net.params.cpu().clone(), acts=stp.acts_online, layout_samples=stp.layout_samples))
    return net, out


def test_td_steps_match_reference_golden_f32(golden):
    """Three full updates (C1 config: B=8, rect clip, gamma .99, lr 1e-4) against goldens produced by the
    reference model class + the reference's process_batch statements + torch.optim.Adam."""
    net, out = _run_steps("f32", 3)
    lr = 1e-4
    tight = []
    for step, o in enumerate(out, start=1):
        np.testing.assert_allclose(o["loss"], float(golden[f"g3_loss_s{step}"]), rtol=1e-3)
        assert relerr(o["q_before"], torch.from_numpy(golden[f"g3_qbefore_s{step}"]).reshape(8, 15)) < 1e-3
        for name, s in net.slots.items():
            if s.kind != 0:
                continue
            g = o["grads"][s.offset:s.offset + s.numel]
            idx = synth.randint(1234, "idx." + name, (min(16, s.numel),), s.numel)
            amax = float(golden[f"g3_gabsmax_s{step}_{name}"])
            ref = golden[f"g3_gsamp_s{step}_{name}"]
            # 1e-3 of the tensor's max on the L2 norm; single sampled elements get 3e-3 because one ReLU whose
            # pre-activation rounds to the other side of 0 (two fp32 implementations) moves a whole weight row
            # by ~1e-3 of max (DESIGN.md "ReLU flips"; measured in profiles/parity_r01.txt)
            # Steps 2 and 3 are a trajectory check, not an op check (3e-2): Adam divides every gradient element by
            # its own magnitude, so elements whose gradient sits at rounding level receive +-lr updates of arbitrary
            # sign in ANY two fp32 implementations (here even run to run: the wgrad split-K sums with f32 atomics);
            # those few weights perturb the next steps' gradients at the 1e-3..1e-2 level.  Step 1 is the strict gate,
            # the Adam kernel itself is gated bit-tight in test_gpu_ops.py::test_adam_matches_torch.
            p = o["params"][s.offset:s.offset + s.numel][idx].numpy()
            pdiff = np.abs(p - golden[f"g3_psamp_s{step}_{name}"])
            if step == 1:
                assert np.abs(g[idx].numpy() - ref).max() <= 3e-3 * amax + 1e-12, (step, name)
                np.testing.assert_allclose(g.double().norm().item(), float(golden[f"g3_gnorm_s{step}_{name}"]), rtol=1e-3)
                # post-Adam parameters: within 2 % of one lr-sized step
                assert pdiff.max() <= 0.02 * lr + 1e-9, (step, name)
            else:
                # trajectory: gradient norms within 5 %; every sampled parameter within one sign flip per step,
                # and (below) 99 % of all sampled parameters still within 2 % of an lr-sized step
                np.testing.assert_allclose(g.double().norm().item(), float(golden[f"g3_gnorm_s{step}_{name}"]), rtol=5e-2)
                assert pdiff.max() <= 2.5 * lr * step, (step, name)
                tight.append(pdiff <= 0.02 * lr * step + 1e-9)
    # (0.97: measured 0.974-0.995 over kernel variants that differ only in fp32 summation order)
    assert np.concatenate(tight).mean() >= 0.97


def l2err(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


class _EngineReLU(torch.nn.Module):
    """Stand-in for a BasicBlock's (shared) nn.ReLU in the ORACLE, test-side only: the first len(masks) calls — the two ReLUs of
    the block in the model(before) pass, the only pass gradients flow through — take the ENGINE's decisions (x * mask, gradient
    g * mask); later calls (the model(after) pass) are plain ReLUs.  With it the float64 oracle differentiates the same piecewise
    linear function the engine evaluated, so what remains between the two gradients is arithmetic, not a ReLU whose
    pre-activation rounds to the other side of zero."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.calls = masks, 0

    def forward(self, x):
        i = self.calls
        self.calls += 1
        return x * self.masks[i].to(x.dtype) if i < len(self.masks) else torch.relu(x)


def _engine_relu_masks(net, acts, layout_samples, n_frames):
    """The engine's ReLU decisions for the first n_frames frames, per BasicBlock the list the oracle's shared nn.ReLU meets in
    the model(before) pass: the reference applies `features` one frame slot at a time (archs/HabitatDQNMultiAction.py:49-51), so
    the block's ReLU is called as (h, o) of slot 0, (h, o) of slot 1, ... — NCHW bool masks of B samples each (engine images
    are sample-major, frame-minor: slot f = images f, f + F, ...)."""
    F = net.num_frames
    out = []
    for b in range(8):
        sp, c = 56 >> (b // 2), 64 << (b // 2)
        full = []
        for name in (f"h{b}", f"o{b}"):
            a = _act(net, acts, layout_samples, name, (layout_samples * F, sp, sp, c))[:n_frames]
            full.append((a.float() > 0).cpu().permute(0, 3, 1, 2).contiguous())
        calls = []
        for f in range(F):
            calls += [full[0][f::F].contiguous(), full[1][f::F].contiguous()]
        out.append(calls)
    return out


def _f64_yardstick(net, engine_grads, make_trainer, tup, what, relu_masks=None):
    """The f32 gradient gate (north_star: 1e-3 relative, fp32): the oracle is run in float64 AND float32 on the same minibatch;
    per gradient tensor the engine's distance to the float64 gradients is measured and compared with the fp32 oracle's own:
        L2:  err(engine, f64) <= max(1e-3, 1.5 * err(oracle_f32, f64))      max element:  <= max(5e-3, 1.5 * err_max(oracle_f32, f64))
    (the rule of tests/test_gpu_basic.py: two fp32 implementations are both measured against the exact answer, neither against
    the other).  With `relu_masks` (the engine's ReLU decisions, `_engine_relu_masks`) a third float64 run differentiates the
    function the engine actually evaluated (`_EngineReLU`); against it the gate is the strict one, L2 <= 1e-3 and max <= 5e-3
    for every tensor.  Returns (bad_natural, bad_forced, report): the tensors outside either gate and a line naming the worst
    tensors with their measured numbers — emitted as a pytest warning so a `-q` log shows how far from 1e-3 the run was."""
    from oracle import ref_cpu

    def run(prec, masks=None):
        tr = make_trainer()
        tr.model.to(prec)
        tr.target_net.to(prec)
        if masks is not None:
            for b in range(8):
                getattr(tr.model.resnet, f"layer{b // 2 + 1}")[b % 2].relu = _EngineReLU(masks[b])
        tr.model.set_train()
        tr.optimizer.zero_grad()
        loss = ref_cpu.process_batch(tr.model, tr.target_net, tr.config, (tup[0].to(prec), tup[1].to(prec)) + tuple(tup[2:]))
        loss.backward()
        return {n: p.grad.double() for n, p in tr.model.named_parameters() if p.grad is not None}
    g32, g64 = run(torch.float32), run(torch.float64)
    g64f = run(torch.float64, relu_masks) if relu_masks is not None else None
    bad, bad_forced, rows, rows_f = [], [], [], []
    for name, r in g64.items():
        s = net.slots[name]
        ge = engine_grads[s.offset:s.offset + s.numel].view(s.shape).double().cpu()
        rn, rm = r.norm().clamp_min(1e-300), r.abs().max().clamp_min(1e-300)
        e_l2, e_mx = ((ge - r).norm() / rn).item(), ((ge - r).abs().max() / rm).item()
        o_l2, o_mx = ((g32[name] - r).norm() / rn).item(), ((g32[name] - r).abs().max() / rm).item()
        rows.append((e_l2, e_mx, o_l2, o_mx, name))
        if e_l2 > max(1e-3, 1.5 * o_l2) or e_mx > max(5e-3, 1.5 * o_mx):
            bad.append((name, e_l2, e_mx, o_l2, o_mx))
        if g64f is not None:
            rf = g64f[name]
            f_l2 = ((ge - rf).norm() / rf.norm().clamp_min(1e-300)).item()
            f_mx = ((ge - rf).abs().max() / rf.abs().max().clamp_min(1e-300)).item()
            rows_f.append((f_l2, f_mx, name))
            if f_l2 > 1e-3 or f_mx > 5e-3:
                bad_forced.append((name, f_l2, f_mx))
    w = max(rows)
    wm = max(rows, key=lambda t: t[1])
    report = (f"{what}: worst gradient tensor vs the float64 oracle: L2 {w[0]:.3g} ({w[4]}; fp32 oracle's own {w[2]:.3g}), "
              f"max element {wm[1]:.3g} ({wm[4]}; fp32 oracle's own {wm[3]:.3g}); gate L2 <= max(1e-3, 1.5 x oracle), "
              f"max <= max(5e-3, 1.5 x oracle): {len(bad)} of {len(rows)} tensors outside")
    if rows_f:
        wf, wfm = max(rows_f), max(rows_f, key=lambda t: t[1])
        report += (f"; vs the float64 oracle on the ENGINE's ReLU decisions: L2 {wf[0]:.3g} ({wf[2]}), max {wfm[1]:.3g} ({wfm[2]}); strict gate "
                   f"1e-3 / 5e-3: {len(bad_forced)} outside")
    return bad, bad_forced, report


@pytest.mark.parametrize("dtype,tol_q,tol_g,batch_seed", [("f32", 1e-3, 1e-3, 101), ("f32", 1e-3, 1e-3, 102), ("f32", 1e-3, 1e-3, 103), ("bf16", 4e-2, None, 101)])
def test_td_step_matches_oracle_all_elements(dtype, tol_q, tol_g, batch_seed):
    """One update compared over every gradient element with the oracle run on the GPU box's host.
    f32 (three minibatches): every gradient tensor is measured against the oracle run in float64 (`_f64_yardstick`; the measured
    worst tensors are emitted as a pytest warning).  Gate 1, always strict: against the float64 oracle that takes the ENGINE's
    ReLU decisions, relative L2 <= 1e-3 (north_star) and max error <= 5e-3 of the tensor's max, every tensor.  Gate 2, against the
    float64 oracle as it is: the same bounds or 1.5x the fp32 oracle's own distance from float64; a ReLU whose pre-activation
    rounds to the other side of zero in the engine (expected about once per 1e7 activations for ANY two fp32 implementations:
    measured 0-1 per minibatch of 13.6 M) moves the tensors downstream of it by ~1e-3 of their norm — a minibatch with such a
    flip is held to 3e-3 / 1.5e-2 there, and the flips themselves to <= 1e-5 of the activations.
    bf16 (throughput mode): the TD error Q_b - y is a difference of O(1) Q-values carrying ~1e-2 bf16 error and
    bf16 activations flip many ReLU masks, so element-wise agreement with an fp32 run is not defined; gate on
    direction and scale instead: the WHOLE gradient must agree with the oracle's to cosine >= 0.995 and 2 % in norm
    (measured: 0.9990 / 0.4 %), every weight tensor to cosine >= 0.97 and 12 % in norm, and the per-channel vectors (BatchNorm
    weights / biases, conv biases: 64-512 elements, each a sum of strongly cancelling terms) to cosine >= 0.95 and 20 % — the
    worst of them, layer1.1.bn1.weight, moves between 9.6 % and 12.1 % when 0.01 % of the upstream bf16 activations round the
    other way (same kernels, different f32 summation order; every other tensor stays within 6.5 %)."""
    from oracle import ref_cpu
    torch.set_num_threads(max(1, torch.get_num_threads()))
    B = 8
    net, out = _run_steps(dtype, 1, B, batch_seed0=batch_seed - 1)

    def make_trainer():
        t = ref_cpu.Trainer(ref_cpu.default_config(), synth.make_state_dict(7))
        t.target_net.load_state_dict(synth.make_state_dict(8))
        return t
    tr = make_trainer()
    (tup, _) = synth.make_batch(batch_seed, B, 1, structured=True, reward_p=0.3)
    d = {}
    loss = tr.step(tup, d)
    assert abs(out[0]["loss"] - loss) <= tol_q * abs(loss) * 5
    assert relerr(out[0]["q_before"], d["before_values"].detach().reshape(B, 15)) < tol_q
    bad = []
    if dtype == "f32":
        m0 = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False)  # pre-update weights (tr.step ran Adam)
        m0.load_state_dict(synth.make_state_dict(7))
        m0.eval()
        feats = _oracle_relu_outputs(m0, tup[0])
        flips = _count_relu_flips(net, out[0]["acts"], out[0]["layout_samples"], B, feats)
        total = sum(int(v.numel()) for v in feats.values())
        assert flips <= 1e-5 * total
        masks = _engine_relu_masks(net, out[0]["acts"], out[0]["layout_samples"], B)
        bad, bad_forced, report = _f64_yardstick(net, out[0]["grads"], make_trainer, tup, f"f32 parity gate (B=8, F=1, minibatch {batch_seed})", masks)
        import warnings  # the warnings summary is what a `pytest -q` log keeps
        warnings.warn(report + f"; ReLU sign disagreements engine vs fp32 oracle: {flips} of {total}")
        assert not bad_forced, bad_forced
        if flips > 0:  # tensors downstream of a flipped ReLU: bounded, not exempted
            bad = [t for t in bad if t[1] > 3e-3 or t[2] > 1.5e-2]
    all_g, all_ref = [], []
    for name, p in tr.model.named_parameters():
        if p.grad is None:
            continue
        s = net.slots[name]
        g = out[0]["grads"][s.offset:s.offset + s.numel].view(s.shape)
        if dtype == "f32":
            continue  # gated above against float64
        else:
            c, ratio = cosine(g, p.grad), (g.double().norm() / p.grad.double().norm()).item()
            small = p.grad.dim() == 1  # per-channel vector
            if c < (0.95 if small else 0.97) or abs(ratio - 1.0) > (0.20 if small else 0.12):
                bad.append((name, c, ratio))
            all_g.append(g.reshape(-1).double().cpu())
            all_ref.append(p.grad.reshape(-1).double())
    assert not bad, bad
    if dtype != "f32":
        G, R = torch.cat(all_g), torch.cat(all_ref)
        assert (torch.dot(G, R) / (G.norm() * R.norm())).item() >= 0.995 and abs((G.norm() / R.norm()).item() - 1.0) <= 0.02
    # frozen resnet.fc untouched; BN statistics untouched
    sd = synth.make_state_dict(7)
    assert torch.equal(net.view("resnet.fc.weight").cpu(), sd["resnet.fc.weight"])
    assert torch.equal(net.view("resnet.bn1.running_var").cpu(), sd["resnet.bn1.running_var"])


def test_target_sync_timing():
    """target_net is refreshed when sample_number % TARGET_UPDATE_INTERVAL == 0, before that step's update
    (train_q_network.py:215-216)."""
    from video_dqn_amd.engine import TDStepper
    B = 2
    net = make_engine("f32", seed=7, max_batch=2 * B)
    stp = TDStepper(net, B, lr=1e-3, gamma=0.99, clip_rect=True, target_update_interval=3)
    (tup, _) = synth.make_batch(5, B, 1, structured=True, reward_p=0.3)
    args = (tup[0].contiguous().to(DEV), tup[1].contiguous().to(DEV), 1, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
    snaps = []
    for step in range(1, 5):
        before_params_packed = stp.packed_target.clone()
        net.pack_weights(net.packed)
        online_packed_fwd_only = net.packed.clone()
        stp.step(*args)
        torch.cuda.synchronize()
        changed = not torch.equal(before_params_packed, stp.packed_target)
        snaps.append((step, changed, torch.equal(stp.packed_target, online_packed_fwd_only)))
    assert [c for _, c, _ in snaps] == [False, False, True, False]
    assert snaps[2][2]  # at step 3 the target equals the online weights *before* step 3's update


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_early_adam_is_the_same_update(dtype):
    """`TDStepper.step` queues Adam for stage 0 / stage 1 behind that stage's gradient unpack (on the gradient stream, under the
    rest of the backward pass) and the remainder at the end; `forward_backward` + `optimizer_step` is one launch at the end
    (loss.backward(); optimizer.step(), train_q_network.py:226-227).  Deterministic mode: the parameters and both moment
    buffers agree bit for bit after three updates."""
    from video_dqn_amd.engine import TDStepper
    B = 4
    (tup, _) = synth.make_batch(5, B, 1, structured=True, reward_p=0.3)
    args = (tup[0].contiguous().to(DEV), tup[1].contiguous().to(DEV), 1, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
    out = []
    for early in (True, False):
        net = make_engine(dtype, seed=7, max_batch=2 * B, deterministic=True)
        stp = TDStepper(net, B, lr=1e-3, gamma=0.99, clip_rect=True, target_update_interval=2)
        for _ in range(3):
            if early:
                stp.step(*args)
            else:
                stp.sample_number += 1
                if stp.sample_number % stp.tui == 0:
                    stp.sync_target()
                stp.forward_backward(*args)
                stp.optimizer_step()
        torch.cuda.synchronize()
        assert stp.adam_step == 3
        out.append((net.params.clone(), stp.exp_avg.clone(), stp.exp_avg_sq.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert not torch.equal(out[0][0], make_engine(dtype, seed=7, max_batch=2 * B).params)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_frames_packed_one_update_ahead_is_the_same_update(dtype):
    """`TDStepper.step(next_frames=...)` packs the NEXT call's frames on the gradient stream during this update
    (vdqn_step_args.packed_frames) — the reference's loop has the next batch a step ahead too (DataLoader prefetch,
    train_q_network.py:213).  Deterministic mode: parameters and loss bit-equal to packing at the start of every update, also when
    an announced batch is NOT the one that arrives (that call packs its own)."""
    from video_dqn_amd.engine import TDStepper
    B = 4
    batches = []
    for seed in (5, 6, 7, 8):
        (tup, _) = synth.make_batch(seed, B, 1, structured=True, reward_p=0.3)
        batches.append((tup[0].contiguous().to(DEV), tup[1].contiguous().to(DEV), 1, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV)))
    out = []
    for mode in ("ahead", "plain", "wrong_announcement"):
        net = make_engine(dtype, seed=7, max_batch=2 * B, deterministic=True)
        stp = TDStepper(net, B, lr=1e-3, gamma=0.99, clip_rect=True, target_update_interval=3)
        losses = []
        for i, b in enumerate(batches):
            nxt = None
            if mode == "ahead" and i + 1 < len(batches):
                nxt = batches[i + 1][:3]
            if mode == "wrong_announcement":
                nxt = batches[(i + 2) % len(batches)][:3]
            losses.append(stp.step(*b, next_frames=nxt).clone())
        torch.cuda.synchronize()
        out.append((net.params.clone(), torch.cat(losses)))
    for k in (1, 2):
        assert torch.equal(out[0][0], out[k][0]) and torch.equal(out[0][1], out[k][1])
    assert stp._packed_bufs[0] is not None  # the wrong announcements were packed (and discarded)


@pytest.mark.parametrize("dtype", ["bf16"])
def test_frames_announced_ahead_may_alias_this_calls_tensors(dtype):
    """ADVICE rounds 4-5: `next_frames` that ARE this call's tensors.  A loop that replays one resident minibatch (bench.py --pack-ahead
    --pool 1) announces them with the explicit promise (fourth element True) and gets a valid pack; a loop that refills fixed staging
    tensors announces without it — the aliased announcement is ignored and the call packs its own frames, whatever wrote the tensors
    (here once through copy_, once through `.data`, which does not advance `_version`).  All bit-equal to the loop without announcements."""
    from video_dqn_amd.engine import TDStepper
    B = 4
    batches = []
    for seed in (5, 6, 7):
        (tup, _) = synth.make_batch(seed, B, 1, structured=True, reward_p=0.3)
        batches.append((tup[0].contiguous().to(DEV), tup[1].contiguous().to(DEV), 1, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV)))

    def run(mode):
        net = make_engine(dtype, seed=7, max_batch=2 * B, deterministic=True)
        stp = TDStepper(net, B, lr=1e-3, gamma=0.99, clip_rect=True, target_update_interval=3)
        sb, sa = torch.empty_like(batches[0][0]), torch.empty_like(batches[0][1])
        losses = []
        for i in range(6):
            b = batches[0] if mode.startswith("replay") else batches[i % 3]
            if mode.startswith("refill"):
                if mode.startswith("refill_data"):
                    sb.data.copy_(b[0]); sa.data.copy_(b[1])
                else:
                    sb.copy_(b[0]); sa.copy_(b[1])
                b = (sb, sa) + b[2:]
            nxt = (b[:3] + ((True,) if mode.startswith("replay") else ())) if mode.endswith("announced") else None
            losses.append(stp.step(*b, next_frames=nxt).clone())
        torch.cuda.synchronize()
        return net.params.clone(), torch.cat(losses)
    for kind in ("replay", "refill", "refill_data"):
        p0, l0 = run(kind)
        p1, l1 = run(kind + "_announced")
        assert torch.equal(p0, p1) and torch.equal(l0, l1), kind


def _act(net, buf, n_samples, name, shape):
    """View of a named activation inside the engine's workspace (vdqn_net_act_offset)."""
    off = net.lib.vdqn_net_act_offset(net.handle, n_samples, name.encode())
    assert off >= 0, name
    tdt = torch.float32 if net.dtype_name == "f32" else torch.bfloat16
    nbytes = int(np.prod(shape)) * (4 if tdt == torch.float32 else 2)
    return buf[off:off + nbytes].view(tdt).view(shape)


def test_forward_activations_layer_by_layer_f32():
    """Every saved activation of the forward (stem, all 8 BasicBlocks, head) against the oracle's module outputs;
    also counts ReLU sign disagreements (flips) — the quantity that bounds gradient parity."""
    from oracle import ref_cpu
    B = 2
    net = make_engine("f32", seed=11)
    (tup, _) = synth.make_batch(23, B, 1, structured=True)
    m = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False)
    m.load_state_dict(synth.make_state_dict(11))
    m.eval()
    feats = {}

    def hook(name):
        return lambda mod, inp, out: feats.__setitem__(name, out.detach())
    m.resnet.relu.register_forward_hook(hook("c1"))
    m.resnet.maxpool.register_forward_hook(hook("pool"))
    for b in range(8):
        blk = getattr(m.resnet, f"layer{b // 2 + 1}")[b % 2]
        blk.register_forward_hook(hook(f"o{b}"))
        blk.bn1.register_forward_hook(hook(f"hpre{b}"))
    m.features[9].register_forward_hook(hook("f8"))
    m.top[1].register_forward_hook(hook("l0"))
    m.top[3].register_forward_hook(hook("l1"))
    with torch.no_grad():
        m(tup[0])
    net.forward(tup[0].contiguous().to(DEV), 1, B)
    torch.cuda.synchronize()
    buf = net._acts[B]
    dims = {"pool": (56, 64)}  # c1 is never materialised: conv1 + max-pool are one kernel
    for b in range(8):
        dims[f"o{b}"] = (56 >> (b // 2), 64 << (b // 2))
        dims[f"h{b}"] = dims[f"o{b}"]
    flips = 0
    for name, (sp, c) in dims.items():
        got = _act(net, buf, B, name, (B, sp, sp, c)).float().cpu().permute(0, 3, 1, 2)
        ref = torch.relu(feats[f"hpre{name[1:]}"]) if name.startswith("h") else feats[name]
        assert relerr(got, ref) < 1e-4, name
        flips += int(((got > 0) != (ref > 0)).sum())
    f8 = _act(net, buf, B, "f8", (B, 5, 5, 64)).float().cpu().permute(0, 3, 1, 2)
    assert relerr(f8, feats["f8"]) < 1e-4
    assert relerr(_act(net, buf, B, "l0", (B, 512)), feats["l0"]) < 1e-4
    assert relerr(_act(net, buf, B, "l1", (B, 256)), feats["l1"]) < 1e-4
    total = sum(B * sp * sp * c for sp, c in dims.values())
    print(f"ReLU sign disagreements: {flips} of {total} activations")
    assert flips <= 1e-5 * total


def _oracle_relu_outputs(model, x_frames):
    """Post-ReLU activations of every trunk layer of the oracle for frames [n,3,224,224]."""
    feats = {}

    def hook(name):
        return lambda mod, inp, out: feats.__setitem__(name, out.detach())
    hs = [model.resnet.maxpool.register_forward_hook(hook("pool"))]  # c1 is fused away; pool > 0 <=> some c1 in the window > 0
    for b in range(8):
        blk = getattr(model.resnet, f"layer{b // 2 + 1}")[b % 2]
        hs.append(blk.register_forward_hook(hook(f"o{b}")))
        hs.append(blk.bn1.register_forward_hook(hook(f"hpre{b}")))
    with torch.no_grad():
        model.features(x_frames)
    for h in hs:
        h.remove()
    for b in range(8):
        feats[f"h{b}"] = torch.relu(feats.pop(f"hpre{b}"))
    return feats


def _count_relu_flips(net, buf, layout_samples, n_frames, feats):
    """ReLU sign disagreements between the engine's saved activations (first n_frames frames) and the oracle."""
    flips = 0
    F = net.num_frames
    total_frames = layout_samples * F
    for name, ref in feats.items():
        sp, c = ref.shape[-1], ref.shape[1]
        got = _act(net, buf, layout_samples, name, (total_frames, sp, sp, c))[:n_frames].float().cpu().permute(0, 3, 1, 2)
        flips += int(((got > 0) != (ref > 0)).sum())
    return flips


def test_td_step_multi_frame_matches_oracle_f32():
    """PANORAMA / PREVIOUS_IMAGES geometry (F = 4 views per sample, top.0 takes 6400 features): one full update.
    Gradients are gated against the oracle run in float64 (`_f64_yardstick`: 1e-3 L2 / 5e-3 max, or 1.5x the fp32 oracle's own
    distance from float64); ReLU sign disagreements with the fp32 oracle are counted and reported, not used as a tolerance."""
    from oracle import ref_cpu
    from video_dqn_amd.engine import NetEngine, TDStepper
    B, F = 3, 4
    net = NetEngine(3, 5, F, True, "f32", 2 * B)
    net.load_tensors(synth.make_state_dict(7, num_frames=F))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    (tup, _) = synth.make_batch(301, B, F, structured=True, reward_p=0.3)
    before, after, act, rew, term, gt, vm = tup
    stp.forward_backward(before.contiguous().to(DEV), after.contiguous().to(DEV), 1, act.to(DEV), rew.float().to(DEV), term.float().to(DEV))
    torch.cuda.synchronize()
    tr = ref_cpu.Trainer(ref_cpu.default_config(), synth.make_state_dict(7, num_frames=F), num_frames=F)
    tr.model.set_train()
    tr.optimizer.zero_grad()
    d = {}
    loss = ref_cpu.process_batch(tr.model, tr.target_net, tr.config, tup, detail=d)
    loss.backward()
    assert abs(stp.loss.item() - loss.item()) <= 1e-4 * abs(loss.item())
    assert relerr(stp.q_before, d["before_values"].detach().reshape(B, 15)) < 1e-3
    feats = _oracle_relu_outputs(tr.model, before.reshape(B * F, 3, 224, 224))
    flips = _count_relu_flips(net, stp.acts_online, stp.layout_samples, B * F, feats)
    assert flips <= 8
    masks = _engine_relu_masks(net, stp.acts_online, stp.layout_samples, B * F)
    bad, bad_forced, report = _f64_yardstick(net, stp.grads, lambda: ref_cpu.Trainer(ref_cpu.default_config(), synth.make_state_dict(7, num_frames=F), num_frames=F),
                                             tup, f"f32 parity gate (B={B}, F={F})", masks)
    import warnings
    warnings.warn(report + f"; ReLU sign disagreements engine vs fp32 oracle: {flips}")
    assert not bad_forced, bad_forced
    if flips > 0:  # with 12 frames a flipped ReLU moves the 7x7-map gradients more: bounded at 1e-2 / 5e-2, not exempted
        bad = [t for t in bad if t[1] > 1e-2 or t[2] > 5e-2]
    assert not bad, bad


def test_side_stream_overlap_matches_serial():
    """The weight gradients / target forward run on a second HIP stream; from the same parameters and batch the flat
    gradient must equal the serialised run up to f32 summation order (a race would show as O(1) differences)."""
    from video_dqn_amd.engine import NetEngine, TDStepper
    B = 16
    net = NetEngine(3, 5, 1, True, "f32", 2 * B)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    for rep in range(6):
        (tup, raw) = synth.make_batch(400 + rep, B, 1, structured=True, reward_p=0.3)
        args = (torch.from_numpy(raw[0]).to(DEV), torch.from_numpy(raw[1]).to(DEV), 0, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
        res = []
        for overlap in (0, 1, 1):
            net.lib.vdqn_net_set_overlap(net.handle, overlap)
            stp.forward_backward(*args)
            torch.cuda.synchronize()
            res.append((stp.grads.clone(), stp.loss.item()))
        for g, l in res[1:]:
            assert abs(l - res[0][1]) <= 1e-6 * abs(res[0][1])
            assert relerr(g, res[0][0]) < 1e-5, rep
    net.lib.vdqn_net_set_overlap(net.handle, 1)


def _act_f32(net, buf, n_samples, name, shape):
    off = net.lib.vdqn_net_act_offset(net.handle, n_samples, name.encode())
    assert off >= 0, name
    return buf[off:off + int(np.prod(shape)) * 4].view(torch.float32).view(shape)


@pytest.mark.parametrize("dtype,B", [("bf16", 16), ("f32", 6)])
def test_deterministic_mode_is_bit_identical_run_to_run(dtype, B):
    """DETERMINISTIC / VDQN_DETERMINISTIC=1 (the reference pins cudnn.deterministic = True, train_q_network.py:88-89): every
    weight gradient goes through the two-stage ordered reduction instead of f32 atomics and the loss is summed by one block,
    so two runs from the same state are BIT-identical over three updates (side stream on: the order in which blocks or
    streams finish must not matter), and agree with the atomic mode to summation-order accuracy."""
    from video_dqn_amd.engine import NetEngine, TDStepper

    def run(det):
        net = NetEngine(3, 5, 1, True, dtype, 2 * B, deterministic=det)
        net.load_tensors(synth.make_state_dict(7))
        stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, target_update_interval=2)
        losses, g1 = [], None
        for step in range(3):
            (tup, raw) = synth.make_batch(700 + step, B, 1, structured=True, reward_p=0.3)
            stp.step(torch.from_numpy(raw[0]).to(DEV), torch.from_numpy(raw[1]).to(DEV), 0, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
            torch.cuda.synchronize()
            losses.append(stp.loss.item())
            if step == 0:
                g1 = stp.grads.clone()
        return net.params.clone(), stp.exp_avg_sq.clone(), losses, g1

    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and torch.equal(a[3], b[3])
    # same arithmetic as the atomic mode, only the summation order differs (first update, before Adam amplifies rounding)
    assert relerr(a[3], c[3]) < (1e-5 if dtype == "f32" else 1e-4)
    assert abs(a[2][0] - c[2][0]) <= 1e-6 * abs(c[2][0])


def test_update_replays_bit_identically_as_a_hip_graph():
    """One update is kernels, stream waits and event records only — no host synchronisation, no allocation inside the library — so
    it can be captured into a hipGraph through the caller's stream (the engine's side streams join the capture through the fork /
    join events they already use), and a replay reproduces the eager update bit for bit (deterministic mode, train_q_network.py:
    88-89; loss.backward() :226 is the captured work).  On this hardware the replay is SLOWER than the stream schedule (7.25 vs
    6.01 ms per update at batch 256, profiles/r04ad_graph_replay_probe.txt), so neither bench.py nor the trainer replays graphs;
    this test keeps the update capturable for callers that need it."""
    from video_dqn_amd.engine import NetEngine, TDStepper
    B = 16
    (tup, raw) = synth.make_batch(812, B, 1, structured=True, reward_p=0.3)
    args = (torch.from_numpy(raw[0]).to(DEV), torch.from_numpy(raw[1]).to(DEV), 0, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
    net = NetEngine(3, 5, 1, True, "bf16", 2 * B, deterministic=True)
    net.load_tensors(synth.make_state_dict(7))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    stp.forward_backward(*args)
    torch.cuda.synchronize()
    g_eager, l_eager = stp.grads.clone(), stp.loss.item()
    assert g_eager.abs().sum().item() > 0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="relaxed"):
        stp.forward_backward(*args)
    for _ in range(2):
        stp.grads.zero_()
        stp.loss.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert stp.loss.item() == l_eager
        assert torch.equal(stp.grads, g_eager)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("bf16", 1e-4)])
def test_default_schedule_gradients_equal_deterministic_per_tensor(dtype, tol):
    """The DEFAULT placement (atomic split-K sums, weight gradients on the side stream, conv1's weight gradient on the caller's
    stream beside block 0's: VDQN_STEM_WGRAD_MAIN, early Adam off here) against deterministic mode (one weight-gradient stream,
    ordered sums) and against the serialised schedule, ONE update, per gradient tensor: relative L2 within f32 summation order.
    The data-gradient chain is the same arithmetic in all three, only the order of the weight-gradient partial sums differs — a
    kernel reading a buffer before its producer on another stream has finished would show as an O(1) tensor error here."""
    from video_dqn_amd.engine import NetEngine, TDStepper
    B = 16
    (tup, raw) = synth.make_batch(811, B, 1, structured=True, reward_p=0.3)
    args = (torch.from_numpy(raw[0]).to(DEV), torch.from_numpy(raw[1]).to(DEV), 0, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))

    def run(det, overlap):
        net = NetEngine(3, 5, 1, True, dtype, 2 * B, deterministic=det)
        net.load_tensors(synth.make_state_dict(7))
        stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
        net.lib.vdqn_net_set_overlap(net.handle, overlap)
        stp.forward_backward(*args)
        torch.cuda.synchronize()
        return net, stp.grads.clone(), stp.loss.item()

    net, g_det, l_det = run(True, 1)
    for rep in range(3):
        for overlap in (1, 0):
            _, g, l = run(False, overlap)
            assert abs(l - l_det) <= 1e-6 * abs(l_det)
            for name, s in net.slots.items():
                if s.kind == 0:
                    assert l2err(g[s.offset:s.offset + s.numel], g_det[s.offset:s.offset + s.numel]) <= tol, (name, overlap, rep)


def test_deterministic_wgrad_operator_matches_atomic_mode():
    """vdqn_conv2d_wgrad with a workspace (ordered two-stage reduction) against the atomic mode, every kernel variant:
    window 64x64 (layer1-3 geometry), generic 128 (layer4, stride 2), generic 64 (1x1 / stride 2), the stem kernel."""
    from video_dqn_amd import ops
    cases = [(6, 14, 64, 64, 3, 1, 1), (5, 7, 512, 512, 3, 1, 1), (4, 14, 128, 256, 3, 2, 1), (4, 14, 128, 256, 1, 2, 0), (3, 9, 256, 256, 3, 1, 1)]
    for n, hi, ci, co, k, stride, pad in cases:
        ho = (hi + 2 * pad - k) // stride + 1
        x = torch.from_numpy(synth.uniform(n * 100 + ci, "x", (n, hi, hi, ci), -1.0, 1.0)).to(torch.bfloat16).to(DEV)
        gy = torch.from_numpy(synth.uniform(n * 100 + co, "gy", (n, ho, ho, co), -1.0, 1.0)).to(torch.bfloat16).to(DEV)
        kw = dict(co=co, r=k, s=k, stride=stride, pad=pad)
        dw_a, db_a = ops.conv2d_wgrad(gy, x, **kw)
        dw_d1, db_d1 = ops.conv2d_wgrad(gy, x, deterministic=True, **kw)
        dw_d2, db_d2 = ops.conv2d_wgrad(gy, x, deterministic=True, **kw)
        torch.cuda.synchronize()
        assert torch.equal(dw_d1, dw_d2) and torch.equal(db_d1, db_d2), (n, hi, ci, co, k)
        assert relerr(dw_d1, dw_a) < 1e-5 and relerr(db_d1, db_a) < 1e-5, (n, hi, ci, co, k)
    # conv1 as the 4x4/1 convolution over the packed frame (stem kernel)
    n = 3
    t_in = torch.from_numpy(synth.uniform(5, "t", (n, 115, 115, 16), -1.0, 1.0)).to(torch.bfloat16).to(DEV)
    gy = torch.from_numpy(synth.uniform(6, "g", (n, 112, 112, 64), -1.0, 1.0)).to(torch.bfloat16).to(DEV)
    kw = dict(co=64, r=4, s=1, stride=1, pad=0, ci=64, pix_stride=16, want_dbias=False)
    a = ops.conv2d_wgrad(gy, t_in, **kw)
    d1 = ops.conv2d_wgrad(gy, t_in, deterministic=True, **kw)
    d2 = ops.conv2d_wgrad(gy, t_in, deterministic=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(d1, d2) and relerr(d1, a) < 1e-5


@pytest.mark.parametrize("env", [
    pytest.param({"VDQN_FUSE_POOL_BWD": "0"}, marks=pytest.mark.variants),      # max-pool backward + stem weight gradient as two launches
    pytest.param({"VDQN_WIN9_BM256": "2"}, marks=pytest.mark.variants),         # 256-row tiles of the nine-tap window kernel everywhere
    pytest.param({"VDQN_WIN9_BM256": "0", "VDQN_WIN9_PERSIST256": "0"}, marks=pytest.mark.variants),  # ... nowhere (rounds 1-4), and their round-4 form: one workgroup per tile
    {"VDQN_FUSE_DS": "1"},            # round 4's default: the 1x1 downsample as its own forward launch (fused in the data gradient only)
    pytest.param({"VDQN_FUSE_DS": "7"}, marks=pytest.mark.variants),            # fused in the forward pass on the generic kernel too (f32 engines)
    {"VDQN_LEAN_EPILOGUE": "0"},      # the window kernels on the shared igemm_epilogue (round 4) instead of the lean ones
    pytest.param({"VDQN_S2WIN_PERSIST": "0"}, marks=pytest.mark.variants),      # stride-2 plane-window kernel: one workgroup per tile (no tile walk)
    pytest.param({"VDQN_S2WIN_PERSIST": "-1", "VDQN_FUSE_DS": "1"}, marks=pytest.mark.variants),  # ... and round 4's kernel for it
    {"VDQN_S2DGRAD_WIN": "0"},        # stride-2 data gradients on the generic class-tiled kernel instead of the plane-window kernel
    pytest.param({"VDQN_WGRAD_TWO_STAGE": "1"}, marks=pytest.mark.variants),    # split-K partials as plain stores + ordered reduce kernels instead of f32 atomics
    pytest.param({"VDQN_WGRAD_WINDOW": "0"}, marks=pytest.mark.variants),       # the 3x3 / stride-1 weight gradients on the generic kernel (no window tiles)
    {"VDQN_WGRAD_STREAMS": "1"},      # all weight gradients on ONE side stream (default: alternating between the two)
    pytest.param({"VDQN_S2WIN": "0"}, marks=pytest.mark.variants),              # stride-2 3x3 forward convolutions on the generic kernel (no plane-window kernel)
    pytest.param({"VDQN_STEM_NOIDX": "0"}, marks=pytest.mark.variants),         # the stem writes the max-pool arg-max bytes of the no-grad frames too
    {"VDQN_EARLY_ADAM": "0"},         # TDStepper.step: one Adam launch behind the whole backward pass
    pytest.param({"VDQN_STEM_WGRAD_MAIN": "0"}, marks=pytest.mark.variants),    # conv1's weight gradient on the side stream behind block 0's instead of beside them
    {"VDQN_SKINNY": "0"},             # the Q-head's layers on the generic tiled kernel instead of the skinny GEMM kernels
    pytest.param({"VDQN_SKINNY_CONV_CFG": "9"}, marks=pytest.mark.variants),    # features.8 on the skinny kernel that reads its input from global memory (default: images in LDS)
    pytest.param({"VDQN_SIDE_PRIORITY": "normal"}, marks=pytest.mark.variants), # the side streams at the caller's stream priority (default: below it)
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) if isinstance(e, dict) else None)
def test_non_default_kernel_selections(env):
    """The switches that select a non-default kernel or stream arrangement (read once per process) keep the engine's parity and
    determinism tests green: re-run them in a child process with the switch set.  Under the driver's plain `-m gpu` the six A/B
    switches that stay in the product run the bf16 all-elements oracle test + the overlap / determinism / early-Adam tests (the
    switches select bf16 kernels and stream placements); with VDQN_TEST_VARIANTS=1 every switch runs, with the f32 float64-yardstick
    case too."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_engine.py"), "-m", "gpu", "-q", "-x",
                        "-k", ("(td_step_matches_oracle_all_elements and 101) or " if os.environ.get("VDQN_TEST_VARIANTS") == "1" else "(td_step_matches_oracle_all_elements and bf16) or ") +
                        "side_stream_overlap or deterministic_mode_is_bit_identical or early_adam_is_the_same"],
                       env=dict(os.environ, **env), cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
