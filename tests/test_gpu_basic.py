"""ARCHITECTURE='basic' (defaults.py:14): train-mode BatchNorm with per-frame-slot batch statistics, average-pool head.
Operator parity against torch.nn.functional.batch_norm, engine parity against the CPU oracle and against the goldens
produced by the reference's own class / process_batch (tests/golden/make_golden_basic.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

from helpers import relerr  # noqa: E402
from video_dqn_amd import synth  # noqa: E402

DEV = "cuda"


def l2err(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("n,hw,c,F,halves", [(6, 14, 256, 1, 1), (8, 7, 512, 2, 2), (12, 56, 64, 3, 2), (4, 28, 128, 4, 1)])
def test_bn_train_fwd_bwd_match_torch(dtype, tol, n, hw, c, F, halves):
    """Per-group batch statistics (group = model call x frame slot), residual + ReLU epilogue, running-stat update order,
    and the backward (dy, dgamma, dbeta) against autograd through F.batch_norm applied group by group."""
    from video_dqn_amd import ops
    seed = n * 1000 + c
    y = torch.from_numpy(synth.uniform(seed, "y", (n, hw, hw, c), -1.0, 2.0)).to(dtype)
    res = torch.from_numpy(synth.uniform(seed, "r", (n, hw, hw, c), -1.0, 1.0)).to(dtype)
    g_out = torch.from_numpy(synth.uniform(seed, "g", (n, hw, hw, c), -1.0, 1.0)).to(dtype)
    gamma = torch.from_numpy(synth.uniform(seed, "ga", (c,), 0.5, 1.5))
    beta = torch.from_numpy(synth.uniform(seed, "be", (c,), -0.5, 0.5))
    rm0 = torch.from_numpy(synth.uniform(seed, "rm", (c,), -0.2, 0.2))
    rv0 = torch.from_numpy(synth.uniform(seed, "rv", (c,), 0.5, 1.5))
    iph = n // halves
    rm, rv = rm0.clone().to(DEV), rv0.clone().to(DEV)
    z, work = ops.bn_train_fwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), rm, rv, resid=res.to(DEV), relu=True, num_frames=F, imgs_per_half=iph)
    # the engine masks the incoming gradient with the ReLU before the BatchNorm backward
    gm = (g_out.to(DEV).float() * (z.float() > 0)).to(dtype)
    dy, dgamma, dbeta = ops.bn_train_bwd(gm, y.to(DEV), work, num_frames=F, imgs_per_half=iph)
    torch.cuda.synchronize()

    # torch reference in f64 on the (dtype-rounded) inputs, group by group in the reference's call order
    yd = y.double().permute(0, 3, 1, 2).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmr, rvr = rm0.double().clone(), rv0.double().clone()
    zr = torch.zeros_like(yd)
    img = torch.arange(n)
    pieces = []
    for half in range(halves):
        for f in range(F):
            sel = img[(img // iph == half) & (img % F == f)]
            o = Fn.batch_norm(yd[sel], rmr, rvr, gd, bd, training=True, momentum=0.1, eps=1e-5)
            pieces.append((sel, o))
    zr = torch.zeros_like(yd)
    for sel, o in pieces:
        zr = zr.index_add(0, sel, o)
    zr = torch.relu(zr + res.double().permute(0, 3, 1, 2))
    assert relerr(z.float().cpu().permute(0, 3, 1, 2), zr.detach()) < tol
    np.testing.assert_allclose(rm.cpu().numpy(), rmr.numpy(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-4, atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), rvr.numpy(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-4, atol=1e-5)
    zr.backward(gm.double().cpu().permute(0, 3, 1, 2))
    assert relerr(dy.float().cpu().permute(0, 3, 1, 2), yd.grad) < max(tol, 1e-4)
    assert relerr(dgamma, gd.grad) < max(tol, 1e-4)
    assert relerr(dbeta, bd.grad) < max(tol, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_train_deterministic_sums(dtype):
    """vdqn_bn_train_fwd / _bwd with a workspace (per-block partial sums added in block order, the workspace handed over full of
    NaN): bit-identical run to run where the atomic mode is not guaranteed to be, and equal to it to summation order.  Many
    blocks per group (48 images of 56 x 56: 147 blocks) so that the order of the cross-block sum matters."""
    from video_dqn_amd import ops
    n, hw, c, F, iph = 48, 56, 64, 2, 24
    y = torch.from_numpy(synth.uniform(9, "y", (n, hw, hw, c), -1.0, 2.0)).to(dtype).to(DEV)
    g = torch.from_numpy(synth.uniform(9, "g", (n, hw, hw, c), -1.0, 1.0)).to(dtype).to(DEV)
    gamma = torch.from_numpy(synth.uniform(9, "ga", (c,), 0.5, 1.5)).to(DEV)
    beta = torch.from_numpy(synth.uniform(9, "be", (c,), -0.5, 0.5)).to(DEV)

    def run(det):
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        z, work = ops.bn_train_fwd(y, gamma, beta, rm, rv, relu=True, num_frames=F, imgs_per_half=iph, deterministic=det)
        w_fwd = work.clone()
        dy, dgamma, dbeta = ops.bn_train_bwd(g, y, work, num_frames=F, imgs_per_half=iph, deterministic=det)
        torch.cuda.synchronize()
        return z, w_fwd, rm, rv, dy, dgamma, dbeta
    a, b, ref = run(True), run(True), run(False)
    for x, y_ in zip(a, b):
        assert torch.isfinite(x.float()).all() and torch.equal(x, y_)
    for x, r in zip(a, ref):
        assert relerr(x, r) < (1e-5 if dtype == torch.float32 else 1e-2)


def test_basic_arch_deterministic_updates_are_bit_identical():
    """DETERMINISTIC covers ARCHITECTURE='basic' (defaults.py:14; train_q_network.py:88-89 pins cudnn.deterministic): with the
    train-mode BatchNorm statistics through the ordered sums, three updates from the same state leave the same bits in
    parameters, running statistics and Adam state, and the first gradient equals the atomic mode's to summation order."""
    from video_dqn_amd.engine import NetEngine, TDStepper
    B, F = 6, 1

    def run(det):
        net = NetEngine(3, 5, F, False, "f32", 2 * B, deterministic=det)
        net.load_tensors(synth.make_state_dict(7, extra_capacity=False, num_frames=F))
        stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True, target_update_interval=2)
        g1 = None
        for step in range(3):
            (tup, raw) = synth.make_batch(720 + step, B, F, structured=True, reward_p=0.3)
            stp.step(torch.from_numpy(raw[0]).to(DEV), torch.from_numpy(raw[1]).to(DEV), 0, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
            torch.cuda.synchronize()
            if step == 0:
                g1 = stp.grads.clone()
        return net.params.clone(), net.bnstats.clone(), stp.exp_avg_sq.clone(), g1
    a, b, c = run(True), run(True), run(False)
    for x, y_ in zip(a, b):
        assert torch.equal(x, y_)
    assert relerr(a[3], c[3]) < 1e-3  # batch statistics amplify summation-order differences of the first update


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_avgpool_fwd_bwd(dtype, tol):
    from video_dqn_amd import ops
    x = torch.relu(torch.from_numpy(synth.uniform(5, "x", (6, 7, 7, 512), -1.0, 1.0))).to(dtype)
    g = torch.from_numpy(synth.uniform(5, "g", (6, 512), -1.0, 1.0)).to(dtype)
    out = ops.avgpool_fwd(x.to(DEV))
    gx = ops.avgpool_bwd(g.to(DEV), x.to(DEV))
    torch.cuda.synchronize()
    assert relerr(out, x.float().mean(dim=(1, 2))) < tol
    ref = (x.float() > 0) * g.float()[:, None, None, :] / 49
    assert relerr(gx, ref) < tol


def make_basic(dtype, seed, F, max_batch, deterministic=None):
    from video_dqn_amd.engine import NetEngine
    net = NetEngine(3, 5, F, False, dtype, max_batch, deterministic=deterministic)
    net.load_tensors(synth.make_state_dict(seed, extra_capacity=False, num_frames=F))
    return net


def test_basic_eval_forward_matches_reference_golden(golden):
    """Eval-mode forward (running statistics folded, average pool, one Linear) against G2's ec0 cases."""
    n = 0
    for ec, pano, B, st in golden["g2_cases"]:
        if ec:
            continue
        F = 4 if pano else 1
        net = make_basic("f32", 11, F, 8)
        (tup, raw) = synth.make_batch(21 + int(B), int(B), F, structured=True)
        ref = torch.from_numpy(golden[f"g2_q_ec0_pano{int(pano)}_B{int(B)}_eval"]).reshape(int(B), 15)
        q = net.forward(tup[0].contiguous().to(DEV), 1, int(B))
        q0 = net.forward(torch.from_numpy(raw[0]).to(DEV), 0, int(B))
        torch.cuda.synchronize()
        assert relerr(q, ref) < 1e-3 and relerr(q0, ref) < 1e-3
        n += 1
    assert n == 2


@pytest.mark.parametrize("F,B", [(1, 5), (4, 3)])
def test_basic_model_train_forward_matches_oracle(F, B):
    """model(x) under model.train(): batch statistics per frame slot, running statistics and num_batches_tracked
    advance exactly like torch's BatchNorm2d; a following eval forward uses the updated statistics."""
    from oracle import ref_cpu
    from video_dqn_amd.model import HabitatDQNMultiAction
    sd = synth.make_state_dict(11, extra_capacity=False, num_frames=F)
    m = HabitatDQNMultiAction(3, 5, extra_capacity=False, panorama=F > 1, num_frames=F, dtype="f32", device=DEV, max_batch=8)
    m.load_state_dict(sd, strict=True)
    ref = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=False, panorama=F > 1, num_frames=F)
    ref.load_state_dict(sd)
    (tup, _) = synth.make_batch(77, B, F, structured=True)
    m.set_train()
    ref.set_train()
    with torch.no_grad():
        for _ in range(2):
            q = m(tup[0])
            qr = ref(tup[0])
            assert q.shape == qr.shape == (B, 5, 3)
            assert relerr(q, qr) < 1e-3
    got, want = m.state_dict(), ref.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k, v in want.items():
        if "num_batches_tracked" in k:
            assert int(got[k]) == int(v) == 2 * F, k
        elif "running_" in k:
            assert relerr(got[k], v) < 1e-4, k
    m.eval()
    ref.eval()
    with torch.no_grad():
        assert relerr(m(tup[0]), ref(tup[0])) < 1e-3
    assert int(m.state_dict()["resnet.bn1.num_batches_tracked"]) == 2 * F  # eval forward does not advance it


def _basic_steps(dtype, F, B, steps, pano, deterministic=None):
    from video_dqn_amd.engine import TDStepper
    net = make_basic(dtype, 7, F, 2 * B, deterministic)
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    tnet = make_basic(dtype, 8, F, 2 * B)
    tnet.pack_weights(stp.packed_target)
    out = []
    for step in range(1, steps + 1):
        (tup, _) = synth.make_batch(400 + 10 * F + step, B, F, structured=True, reward_p=0.3)
        before, after, act, rew, term, gt, vm = tup
        loss = stp.step(before.contiguous().to(DEV), after.contiguous().to(DEV), 1, act.to(DEV), rew.float().to(DEV), term.float().to(DEV))
        torch.cuda.synchronize()
        out.append(dict(loss=loss.item(), q_before=stp.q_before.cpu().clone(), grads=stp.grads.cpu().clone(),
                        params=net.params.cpu().clone(), bnstats=net.bnstats.cpu().clone(), nbt=net.num_batches_tracked.cpu().clone()))
    return net, out


@pytest.mark.parametrize("tag,pano,F,B,steps", [("F1", False, 1, 6, 2), ("F4", True, 4, 3, 1)])
def test_basic_td_steps_match_reference_golden_f32(golden_basic, tag, pano, F, B, steps):
    """Full updates of ARCHITECTURE='basic' against G5 (reference class + reference process_batch + torch Adam): loss, Q(s),
    running statistics, num_batches_tracked at 1e-3; gradients and post-Adam parameters as follows.

    With batch-statistics BatchNorm a ReLU whose pre-activation rounds to the other side of zero changes the statistics of
    its whole group, so two fp32 implementations differ by more than 1e-3 in some gradient tensors — the reference's own
    fp32 run is 1.8e-2 .. 3.9e-2 (max-error) away from the same code run in float64 (stored in the golden file by
    make_golden_basic.py as ref32_vs_ref64_worst_*).  The gate therefore measures the engine against the reference's
    float64 gradients and requires it to be as close as 1e-3/5e-3 or as close as 1.5x the reference's own fp32 run."""
    g = golden_basic
    # (deterministic mode, as the reference runs — cudnn.deterministic = True, train_q_network.py:88-89: the atomic-sum mode's f32
    # result wanders run to run by as much as the gate's margin, see test_basic_td_step_all_elements_vs_oracle_f32)
    net, out = _basic_steps("f32", F, B, steps, pano, deterministic=True)
    lr = 1e-4
    e_max, e_l2 = float(g[f"g5_{tag}_s1_ref32_vs_ref64_worst_max"]), float(g[f"g5_{tag}_s1_ref32_vs_ref64_worst_l2"])
    tol_elem, tol_norm = max(5e-3, 1.5 * e_max), max(1e-3, 1.5 * e_l2)
    for step, o in enumerate(out, start=1):
        k = f"g5_{tag}_s{step}"
        np.testing.assert_allclose(o["loss"], float(g[f"{k}_loss"]), rtol=1e-3)
        assert relerr(o["q_before"], torch.from_numpy(g[f"{k}_qbefore"]).reshape(B, 15)) < 1e-3
        assert int(o["nbt"][0]) == int(o["nbt"][-1]) == 2 * F * step
        tight = []
        for name, s in net.slots.items():
            if s.kind in (2, 3):
                got = o["bnstats"][s.offset:s.offset + s.numel]
                assert relerr(got, torch.from_numpy(g[f"{k}_bn_{name}"])) < 1e-3, (step, name)
            if s.kind != 0:
                continue
            gr = o["grads"][s.offset:s.offset + s.numel]
            idx = synth.randint(1234, "idx." + name, (min(16, s.numel),), s.numel)
            pdiff = np.abs(o["params"][s.offset:s.offset + s.numel][idx].numpy() - g[f"{k}_psamp_{name}"])
            if step == 1:
                amax = float(g[f"{k}_gabsmax64_{name}"])
                assert np.abs(gr[idx].double().numpy() - g[f"{k}_gsamp64_{name}"]).max() <= tol_elem * amax + 1e-12, (step, name)
                np.testing.assert_allclose(gr.double().norm().item(), float(g[f"{k}_gnorm64_{name}"]), rtol=tol_norm, err_msg=name)
                assert pdiff.max() <= 2.5 * lr, (step, name)  # first Adam step = lr * sign(g): at most one sign flip
                tight.append(pdiff <= 0.02 * lr + 1e-9)
            else:  # trajectory check (see test_gpu_engine.py::test_td_steps_match_reference_golden_f32)
                np.testing.assert_allclose(gr.double().norm().item(), float(g[f"{k}_gnorm_{name}"]), rtol=5e-2, err_msg=name)
                assert pdiff.max() <= 2.5 * lr * step, (step, name)
        if step == 1:
            assert np.concatenate(tight).mean() >= 0.99


@pytest.mark.parametrize("F,B", [(1, 6), (4, 3), (1, 32)])
def test_basic_td_step_all_elements_vs_oracle_f32(F, B):
    """Every gradient element of one update against the oracle run live in float64 and float32 on the host: the engine's
    distance to the float64 gradients must be within 1e-3 (L2) / 5e-3 (max) or within 1.5x the fp32 oracle's own distance.

    Run in DETERMINISTIC mode, as the reference runs (cudnn.deterministic = True, train_q_network.py:88-89): with batch
    statistics the f32 result is chaotic in the summation order — over seven runs of the atomic-sum mode the worst element's
    error at F = 4 was 1.59e-2 .. 2.73e-2 and the worst tensor's L2 2.8e-3 .. 6.3e-3 (profiles/r03y_basic_f32_run_to_run.txt)
    against 1.82e-2 / 6.0e-3 for the oracle's own fp32 run, i.e. one draw in seven landed 0.3 % above the 1.5x line.  The ordered
    sums give one reproducible number (1.60e-2 / 6.0e-3 at F = 4).  The atomic-sum mode is held to the deterministic run below."""
    from oracle import ref_cpu
    net, out = _basic_steps("f32", F, B, 1, F > 1, deterministic=True)
    _, out_atomic = _basic_steps("f32", F, B, 1, F > 1)
    cfg = ref_cpu.default_config(ARCHITECTURE="basic", PANORAMA=F > 1)
    (tup, _) = synth.make_batch(400 + 10 * F + 1, B, F, structured=True, reward_p=0.3)
    grads = {}
    for prec in (torch.float32, torch.float64):
        tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7, extra_capacity=False, num_frames=F), num_frames=F)
        tr.target_net.load_state_dict(synth.make_state_dict(8, extra_capacity=False, num_frames=F))
        tr.model.to(prec)
        tr.target_net.to(prec)
        tr.model.set_train()
        loss = ref_cpu.process_batch(tr.model, tr.target_net, cfg, (tup[0].to(prec), tup[1].to(prec)) + tuple(tup[2:]))
        loss.backward()
        grads[prec] = {n: p.grad.double() for n, p in tr.model.named_parameters() if p.grad is not None}
        assert abs(out[0]["loss"] - loss.item()) <= 1e-4 * abs(loss.item())
    worst = dict(eng_max=0.0, eng_l2=0.0, ref_max=0.0, ref_l2=0.0)
    for name, r in grads[torch.float64].items():
        s = net.slots[name]
        ge = out[0]["grads"][s.offset:s.offset + s.numel].view(s.shape).double()
        g32 = grads[torch.float32][name]
        worst["eng_max"] = max(worst["eng_max"], ((ge - r).abs().max() / r.abs().max()).item())
        worst["eng_l2"] = max(worst["eng_l2"], ((ge - r).norm() / r.norm()).item())
        worst["ref_max"] = max(worst["ref_max"], ((g32 - r).abs().max() / r.abs().max()).item())
        worst["ref_l2"] = max(worst["ref_l2"], ((g32 - r).norm() / r.norm()).item())
    print("worst gradient error vs float64:", worst)
    import warnings  # (B = 32: does the chaos of batch statistics condition with the batch?  the numbers land in the -q log)
    warnings.warn(f"basic f32 gate F={F} B={B}: engine vs float64 max {worst['eng_max']:.3g} L2 {worst['eng_l2']:.3g}; fp32 oracle's own "
                  f"max {worst['ref_max']:.3g} L2 {worst['ref_l2']:.3g}; plain 1e-3 L2 line {'holds' if worst['eng_l2'] <= 1e-3 else 'does not hold'}")
    assert worst["eng_max"] <= max(5e-3, 1.5 * worst["ref_max"]), worst
    assert worst["eng_l2"] <= max(1e-3, 1.5 * worst["ref_l2"]), worst
    # The default (atomic-sum) mode is held to the DETERMINISTIC engine, not to float64: same loss, and the whole gradient within
    # 1e-3 (relative L2 and relative max element) of the deterministic run.  Its distance to float64 is a draw from a chaotic map —
    # at B = 32, 4 of 20 identical runs land on one DISCRETE other outcome (the same element, layer4.1.conv2.weight[672735], at
    # 2.5644e-2 every time, worst tensor L2 4.3-4.6e-3, against 0.9-1.7e-2 / 1.1-2.2e-3 for the other 16 and 1.07e-2 / 1.5e-3 for
    # the fp32 oracle itself), with and without the side streams (2 of 20 on one stream): a decision flipped by the order of the
    # batch-statistic sums, not a race — while the distance to the deterministic run stays at 3e-5 .. 1.6e-4 in all 40 runs
    # (profiles/r04ay_basic_b32_atomic_mode_spread.txt).  A kernel reading a buffer before its producer finished shows as O(0.1-1)
    # here.  The float64 numbers of this draw go to the log.
    assert abs(out_atomic[0]["loss"] - out[0]["loss"]) <= 1e-5 * abs(out[0]["loss"])
    wa = dict(max=0.0, l2=0.0)
    for name, r in grads[torch.float64].items():
        s = net.slots[name]
        ge = out_atomic[0]["grads"][s.offset:s.offset + s.numel].view(s.shape).double()
        wa["max"] = max(wa["max"], ((ge - r).abs().max() / r.abs().max()).item())
        wa["l2"] = max(wa["l2"], ((ge - r).norm() / r.norm()).item())
    g_det, g_atm = out[0]["grads"].double(), out_atomic[0]["grads"].double()
    d_l2 = ((g_atm - g_det).norm() / g_det.norm()).item()
    d_max = ((g_atm - g_det).abs().max() / g_det.abs().max()).item()
    print("atomic-sum mode, worst gradient error vs float64:", wa, "distance to the deterministic run:", d_l2, d_max)
    warnings.warn(f"basic f32 F={F} B={B}, atomic-sum mode: vs float64 max {wa['max']:.3g} L2 {wa['l2']:.3g}; vs deterministic run L2 {d_l2:.3g} max {d_max:.3g}")
    assert d_l2 <= 1e-3 and d_max <= 1e-3, (d_l2, d_max, wa, worst)
    # ... and, loosely, to float64 itself (a fault that hit both modes alike would pass the line above): within 4x the
    # fp32 oracle's own distance — the discrete outlier of the chaotic draw sits at 2.4x (max) / 3.1x (L2) of it, a kernel that
    # reads a buffer too early at 10-100x
    assert wa["max"] <= max(5e-3, 4.0 * worst["ref_max"]), (wa, worst)
    assert wa["l2"] <= max(1e-3, 4.0 * worst["ref_l2"]), (wa, worst)


def test_basic_td_step_bf16_direction_and_scale():
    """bf16 throughput mode of the basic arch against the fp32 oracle: cosine / norm-ratio gate per tensor."""
    from oracle import ref_cpu
    B, F = 8, 1
    net, out = _basic_steps("bf16", F, B, 1, False)
    cfg = ref_cpu.default_config(ARCHITECTURE="basic")
    tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7, extra_capacity=False))
    tr.target_net.load_state_dict(synth.make_state_dict(8, extra_capacity=False))
    (tup, _) = synth.make_batch(400 + 10 * F + 1, B, F, structured=True, reward_p=0.3)
    d = {}
    loss = tr.step(tup, d)
    assert abs(out[0]["loss"] - loss) <= 0.2 * abs(loss)
    assert relerr(out[0]["q_before"], d["before_values"].detach().reshape(B, 15)) < 6e-2
    bad = []
    for name, p in tr.model.named_parameters():
        if p.grad is None:
            continue
        s = net.slots[name]
        gr = out[0]["grads"][s.offset:s.offset + s.numel].view(s.shape)
        c, ratio = cosine(gr, p.grad), (gr.double().norm() / p.grad.double().norm()).item()
        if c < 0.95 or abs(ratio - 1.0) > 0.15:
            bad.append((name, c, ratio))
    assert not bad, bad
    for name, s in net.slots.items():
        if s.kind in (2, 3):
            assert relerr(out[0]["bnstats"][s.offset:s.offset + s.numel], tr.model.state_dict()[name]) < 3e-2, name
