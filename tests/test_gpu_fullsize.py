"""The benchmarked launch geometries under test (BASELINE configs 2 and 5): one TD update at batch 256 (bf16) and at 12 views
x batch 16 through the real kernels' full-size grids — properties that need no oracle at this size: finite loss, the
side-stream overlap changes nothing (bit-equal in deterministic mode, summation-order-equal otherwise), and the bf16
gradient points where the f32 engine's gradient points."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

from video_dqn_amd import synth  # noqa: E402

DEV = "cuda"


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300)).item()


def _inputs(B, F, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    before = torch.randint(0, 256, (B, F, 224, 224, 3), dtype=torch.uint8, device=DEV, generator=g)
    after = torch.randint(0, 256, (B, F, 224, 224, 3), dtype=torch.uint8, device=DEV, generator=g)
    # structure on top of the noise so that Q-values differ between samples: a per-sample brightness offset
    off = torch.randint(0, 120, (B, 1, 1, 1, 1), dtype=torch.int16, device=DEV, generator=g)
    before = ((before.to(torch.int16) >> 1) + off).clamp(0, 255).to(torch.uint8)
    after = ((after.to(torch.int16) >> 1) + off).clamp(0, 255).to(torch.uint8)
    act = torch.randint(0, 3, (B,), dtype=torch.int64, device=DEV, generator=g)
    rew = (torch.rand((B, 5), device=DEV, generator=g) < 0.3).float()
    return before, after, act, rew


def _grads(dtype, B, F, inputs, overlap, deterministic):
    from video_dqn_amd.engine import NetEngine, TDStepper
    net = NetEngine(3, 5, F, True, dtype, 2 * B, deterministic=deterministic)
    net.load_tensors(synth.make_state_dict(7, num_frames=F))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    net.lib.vdqn_net_set_overlap(net.handle, overlap)
    before, after, act, rew = inputs
    stp.forward_backward(before, after, 0, act, rew, rew)
    torch.cuda.synchronize()
    out = (stp.grads.clone(), stp.loss.item(), stp.q_before.clone())
    del stp, net
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("B,F", [(256, 1), (16, 12)], ids=["C2_batch256", "C5_12views_batch16"])
def test_full_size_step_properties(B, F):
    inputs = _inputs(B, F, 99 + F)
    g_det_ov, loss_ov, q_ov = _grads("bf16", B, F, inputs, overlap=1, deterministic=True)
    g_det_se, loss_se, q_se = _grads("bf16", B, F, inputs, overlap=0, deterministic=True)
    assert torch.isfinite(g_det_ov).all() and loss_ov == loss_ov and 0.0 < loss_ov < 10.0
    # deterministic mode: the second stream only changes WHEN kernels run, never what they compute
    assert torch.equal(g_det_ov, g_det_se) and loss_ov == loss_se and torch.equal(q_ov, q_se)
    # default (atomic) mode: the same gradient up to f32 summation order of the split weight-gradient tiles
    g_at, loss_at, _ = _grads("bf16", B, F, inputs, overlap=1, deterministic=False)
    c_at = cosine(g_at, g_det_ov)
    assert c_at >= 0.9999 and abs(loss_at - loss_ov) <= 1e-6 * abs(loss_ov)
    # against the f32 engine (exact-f32 MFMA, the parity mode) on the same minibatch
    g32, loss32, q32 = _grads("f32", B, F, inputs, overlap=1, deterministic=True)
    c32 = cosine(g_det_ov, g32)
    nr = (g_det_ov.double().norm() / g32.double().norm()).item()
    qerr = ((q_ov - q32).abs().max() / q32.abs().max()).item()
    warnings.warn(f"full-size step B={B} F={F}: loss bf16 {loss_ov:.6f} / f32 {loss32:.6f}; gradient cosine bf16 vs f32 {c32:.5f}, norm ratio {nr:.4f}; "
                  f"Q(s) max rel err {qerr:.2e}; atomic vs deterministic cosine {c_at:.7f}")
    assert c32 >= 0.99 and abs(nr - 1.0) <= 0.05
    assert qerr < 4e-2 and abs(loss_ov - loss32) <= 5e-2 * abs(loss32)


# ---------------------------------------------------------------------------------------------------------------------------
# The benchmarked geometries against the ORACLE (oracle/ref_cpu.py run on the GPU box's host), not against the engine itself:
# the persistent / multi-round / XCD-remapped launch shapes that bench.py times only exist at these sizes.
# ---------------------------------------------------------------------------------------------------------------------------
class _GraphForFirstCallOnly(torch.nn.Module):
    """Test-side memory economy around an ORACLE module: process_batch (train_q_network.py:131-156) differentiates model(before)
    only — target_net(after) is `.detach()`ed and model(after) feeds an argmax — so the later calls run under no_grad, in chunks.
    Values are identical to the reference's graph-building calls; ~2/3 of the host memory (tens of GB in float64 at batch 256)
    is not allocated."""

    def __init__(self, m, graph_calls):
        super().__init__()
        self.m, self.graph_calls, self.calls = m, graph_calls, 0

    def set_train(self):
        self.m.set_train()

    def forward(self, x):
        self.calls += 1
        if self.calls <= self.graph_calls:
            return self.m(x)
        with torch.no_grad():
            return torch.cat([self.m(x[i:i + 32]) for i in range(0, x.shape[0], 32)], 0)


def _oracle_run(F, tup, prec, wrap=None, wrap_target=None):
    """loss, Q(s), per-parameter gradients of one process_batch + backward of the oracle in precision `prec`.
    wrap(model) / wrap_target(target_net) -> the callables that stand for the two networks (the bf16-emulating forms); default:
    the oracle modules themselves."""
    from oracle import ref_cpu
    cfg = ref_cpu.default_config()
    tr = ref_cpu.Trainer(cfg, synth.make_state_dict(7, num_frames=F), num_frames=F)
    tr.target_net.load_state_dict(synth.make_state_dict(8, num_frames=F))
    tr.model.to(prec)
    tr.target_net.to(prec)
    tr.model.set_train()
    tr.optimizer.zero_grad()
    model = wrap(tr.model) if wrap is not None else _GraphForFirstCallOnly(tr.model, 1)
    target = _GraphForFirstCallOnly(wrap_target(tr.target_net) if wrap_target is not None else tr.target_net, 0)
    d = {}
    batch = (tup[0].to(prec), tup[1].to(prec)) + tuple(tup[2:])
    loss = ref_cpu.process_batch(model, target, cfg, batch, detail=d)
    loss.backward()
    grads = {n: p.grad.double() for n, p in tr.model.named_parameters() if p.grad is not None}
    return loss.item(), d["before_values"].detach().double().reshape(-1, 15), grads, tr


def _engine_run(dtype, B, F, tup, deterministic):
    from video_dqn_amd.engine import NetEngine, TDStepper
    net = NetEngine(3, 5, F, True, dtype, 2 * B, deterministic=deterministic)
    net.load_tensors(synth.make_state_dict(7, num_frames=F))
    stp = TDStepper(net, B, lr=1e-4, gamma=0.99, clip_rect=True)
    tnet = NetEngine(3, 5, F, True, dtype, 2 * B)
    tnet.load_tensors(synth.make_state_dict(8, num_frames=F))
    tnet.pack_weights(stp.packed_target)
    del tnet
    stp.forward_backward(tup[0].contiguous().to(DEV), tup[1].contiguous().to(DEV), 1, tup[2].to(DEV), tup[3].float().to(DEV), tup[4].float().to(DEV))
    torch.cuda.synchronize()
    return net, stp


def _tensor_errors(net, engine_grads, ref):
    """per gradient tensor: (relative L2 error, max error / max |ref|, name) of the engine against `ref` (name -> f64 tensor)"""
    rows = []
    for name, r in ref.items():
        s = net.slots[name]
        ge = engine_grads[s.offset:s.offset + s.numel].view(s.shape).double().cpu()
        rows.append((((ge - r).norm() / r.norm().clamp_min(1e-300)).item(), ((ge - r).abs().max() / r.abs().max().clamp_min(1e-300)).item(), name))
    return rows


def _flat(net, grads):
    return torch.cat([grads[n].reshape(-1) for n, s in net.slots.items() if s.kind == 0 and n in grads])


def _engine_flat(net, engine_grads, ref):
    return torch.cat([engine_grads[s.offset:s.offset + s.numel].double().cpu() for n, s in net.slots.items() if s.kind == 0 and n in ref])


def _engine_head_masks(net, stp, B, F):
    """The engine's ReLU decisions of the head in the model(before) pass: features.8's output per frame slot (NCHW bool), the two
    hidden layers of `top` — the stored activations > 0, exactly what its data-gradient kernels mask with."""
    from test_gpu_engine import _act
    L, n = stp.layout_samples, B * F
    f8 = _act(net, stp.acts_online, L, "f8", (L * F, 5, 5, 64))[:n].float().cpu().view(B, F, 5, 5, 64).permute(1, 0, 4, 2, 3) > 0
    return {"f8": [f8[f].contiguous() for f in range(F)],
            "l0": _act(net, stp.acts_online, L, "l0", (L, 512))[:B].float().cpu() > 0,
            "l1": _act(net, stp.acts_online, L, "l1", (L, 256))[:B].float().cpu() > 0}


def _bf16_emulation_check(B, F, tup, net, stp, failures, notes):
    """The bf16 engine's update (`stp`, after forward_backward on `tup`) against the bf16-emulating oracle run with the engine's
    ReLU masks (blocks AND head): stored activations layer by layer (2e-2 of the tensor's max), gradients per tensor, loss.

    Gradient gate: relative L2 <= 1.2e-2 per tensor at batch 256, 1.5e-2 at the small batches.  Measured (profiles/r04d_*): worst
    tensor 9.1e-3 at B=256 (conv1.weight), 9.8e-3 at 12 views x 16, 9.9e-3 / 1.12e-2 at B=3 x 4 views / B=8 — against 3e-2 .. 1.4e-1 for
    the same tensors before the head's masks were forced, and 7e-3 whole-gradient for the plain fp32 oracle on the same masks.  1e-2
    is the floor of this instrument, not of the kernels: the emulation reproduces the engine's stem output to 9e-6, but a different
    f32 summation order flips a bf16 rounding in ~1e-3 of the elements, the flips compound layer by layer (4e-5, 2e-4, 1e-3, ... 5e-3
    relative L2 at layer4: profiles/r04d_diag_bf16_emulation_b8.txt), and the gradients inherit that 0.5-1 % (the stored gradients
    agree to 4e-3 at the head and 9e-3 at layer1)."""
    import bf16_emulation as emu
    from test_gpu_engine import _act, _engine_relu_masks
    n = B * F
    eg = stp.grads.cpu()
    masks = _engine_relu_masks(net, stp.acts_online, stp.layout_samples, n)
    head_masks = _engine_head_masks(net, stp, B, F)
    rec = {}
    # BOTH networks emulated: the TD error Q_b - y of a no-reward entry is a small difference of Q-values, so a target pass in plain
    # fp32 (1e-2-level off the engine's bf16 target Q) would move dL/dQ itself by several per cent
    loss_e, q_e, g_e, _ = _oracle_run(F, tup, torch.float32, wrap=lambda m: emu.EmulatedNet(m, masks, rec, graph_first_call_only=True, head_masks=head_masks),
                                      wrap_target=lambda t: emu.EmulatedNet(t))
    del masks
    L = stp.layout_samples
    worst_act = (0.0, "")
    for name, lst in rec.items():
        if name in ("l0", "l1", "q"):
            ref = lst
            got = (_act(net, stp.acts_online, L, name, (L, ref.shape[1]))[:B].float().cpu() if name != "q"
                   else stp.q_before.cpu())
        else:
            c, sp = lst[0].shape[1], lst[0].shape[2]
            a = _act(net, stp.acts_online, L, name, (L * F, sp, sp, c))[:n].float().cpu()
            got = a.view(B, F, sp, sp, c).permute(1, 0, 4, 2, 3)          # [slot][B, C, H, W]
            ref = torch.stack(lst, 0)
        e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        worst_act = max(worst_act, (e, name))
        if e > 2e-2:
            failures.append(("bf16 activation vs emulating oracle", name, e))
    rows = _tensor_errors(net, eg, g_e)
    w = max(rows)
    gate = 1.2e-2 if B >= 256 else 1.5e-2
    over = [(nm, l2) for l2, _, nm in rows if l2 > gate]
    notes.append(f"bf16 engine vs bf16-emulating oracle (engine's ReLU masks): loss {stp.loss.item():.6f} / {loss_e:.6f}, worst stored activation "
                 f"{worst_act[0]:.3g} of its tensor's max ({worst_act[1]}), worst gradient tensor L2 {w[0]:.3g} ({w[2]}), "
                 f"{sum(1 for l2, _, _ in rows if l2 > 1e-2)} of {len(rows)} over 1e-2, {len(over)} over the gate {gate:g}")
    if over:
        failures.append((f"bf16 gradient tensors vs emulating oracle over {gate:g}", sorted(over, key=lambda t: -t[1])[:8]))
    if abs(stp.loss.item() - loss_e) > 5e-3 * abs(loss_e):
        failures.append(("bf16 loss vs emulating oracle", stp.loss.item(), loss_e))


# (C5: four oracle passes over 576 frames, two of them in float64 = five minutes of host time for one test.  Under the driver's plain
# `-m gpu` the 12-view geometry is covered by test_full_size_step_properties[C5] (f32 engine vs bf16 engine, overlap, determinism) and
# by the F = 4 oracle tests; this case runs with VDQN_TEST_SLOW=1 — tools/job.sh slow — and its log is committed under profiles/)
@pytest.mark.parametrize("B,F,seed", [(256, 1, 501), pytest.param(16, 12, 502, marks=pytest.mark.slow_oracle)], ids=["C2_batch256", "C5_12views_batch16"])
def test_full_size_step_vs_oracle(B, F, seed):
    """One TD update at the benchmarked geometry (train_q_network.py:126-181,226), engine against oracle on the same minibatch:

    f32 engine (deterministic sums) — loss and Q(s) at 1e-3; every gradient tensor under the NATURAL gate
        err(engine, oracle_f64) <= max(1e-3, 1.5 * err(oracle_f32, oracle_f64))  (L2; max element: 5e-3 likewise)
      with the oracle run as the reference would run it (its own ReLU decisions) and NO relaxed branch: at 256 samples a ReLU
      whose pre-activation rounds to the other side of zero is one of ~4e8 and must drown; the sign disagreements are counted
      and printed.
    bf16 engine (the benchmarked kernels, default atomic mode) — against the SAME fp32 oracle: whole-gradient cosine >= 0.999,
      norm within 2 %, per tensor as tests/test_gpu_engine.py (cosine 0.97 / 0.95 for per-channel vectors); and against the
      bf16-EMULATING oracle (tests/bf16_emulation.py: the oracle with the engine's rounding points and the engine's ReLU masks):
      every stored activation layer by layer within 2e-2 of the tensor's max, every gradient tensor within 1.2e-2 relative L2
      (`_bf16_emulation_check`: measured 9.1e-3, and why 1e-2 is this instrument's floor)."""
    import time
    from test_gpu_engine import _count_relu_flips, _oracle_relu_outputs
    t0 = time.time()
    (tup, _) = synth.make_batch(seed, B, F, structured=True, reward_p=0.3)
    n = B * F
    failures, notes = [], []

    # ---- f32 engine vs the oracle in float32 / float64 --------------------------------------------------------------------
    loss32, q32, g32, tr32 = _oracle_run(F, tup, torch.float32)
    loss64, q64, g64, _ = _oracle_run(F, tup, torch.float64)
    net, stp = _engine_run("f32", B, F, tup, deterministic=True)
    eg = stp.grads.cpu()
    if abs(stp.loss.item() - loss64) > 1e-3 * abs(loss64):
        failures.append(("f32 loss", stp.loss.item(), loss64))
    qerr = ((stp.q_before.double().cpu() - q64).abs().max() / q64.abs().max()).item()
    if qerr > 1e-3:
        failures.append(("f32 Q(s)", qerr))
    feats = _oracle_relu_outputs(tr32.model, tup[0].reshape(n, 3, 224, 224))
    flips = _count_relu_flips(net, stp.acts_online, stp.layout_samples, n, feats)
    total = sum(int(v.numel()) for v in feats.values())
    del feats
    rows = _tensor_errors(net, eg, g64)
    own = {nm: (((g32[nm] - r).norm() / r.norm().clamp_min(1e-300)).item(), ((g32[nm] - r).abs().max() / r.abs().max().clamp_min(1e-300)).item())
           for nm, r in g64.items()}
    outside = [(nm, l2, mx, own[nm]) for l2, mx, nm in rows if l2 > max(1e-3, 1.5 * own[nm][0]) or mx > max(5e-3, 1.5 * own[nm][1])]
    w, wm = max(rows), max(rows, key=lambda t: t[1])
    notes.append(f"f32 engine vs float64 oracle: loss {stp.loss.item():.7f} / {loss64:.7f}, Q(s) max rel err {qerr:.2e}; worst gradient tensor L2 "
                 f"{w[0]:.3g} ({w[2]}; fp32 oracle's own {own[w[2]][0]:.3g}), max element {wm[1]:.3g} ({wm[2]}; fp32 oracle's own {own[wm[2]][1]:.3g}); "
                 f"natural gate max(1e-3, 1.5 x oracle) / max(5e-3, 1.5 x oracle): {len(outside)} of {len(rows)} tensors outside; ReLU sign "
                 f"disagreements engine vs fp32 oracle: {flips} of {total}")
    if B >= 256:
        if outside:  # 256 samples: a flipped ReLU is one of ~4e8 decisions and drowns — no relaxed branch
            failures.append(("f32 gradient tensors outside the natural gate", outside[:6]))
    else:
        # 16 samples (192 frames): measured — the few dozen ReLU sign disagreements do NOT drown (14 of 68 tensors land between 1e-3
        # and 1.7e-3, profiles/r04a_*): a finding, reported above.  Held instead to (1) the STRICT gate against the float64 oracle
        # that differentiates the function the engine evaluated (its ReLU decisions: 1e-3 L2 / 5e-3 max, every tensor) and (2) the
        # natural comparison bounded at 3e-3 / 1.5e-2, as tests/test_gpu_engine.py holds the 8-sample minibatches
        from test_gpu_engine import _EngineReLU, _engine_relu_masks
        masks = _engine_relu_masks(net, stp.acts_online, stp.layout_samples, n)

        def forced(m):
            for b in range(8):
                getattr(m.resnet, f"layer{b // 2 + 1}")[b % 2].relu = _EngineReLU(masks[b])
            return _GraphForFirstCallOnly(m, 1)
        _, _, g64f, _ = _oracle_run(F, tup, torch.float64, wrap=forced)
        del masks
        rows_f = _tensor_errors(net, eg, g64f)
        wf, wfm = max(rows_f), max(rows_f, key=lambda t: t[1])
        bad_forced = [(nm, l2, mx) for l2, mx, nm in rows_f if l2 > 1e-3 or mx > 5e-3]
        notes.append(f"vs the float64 oracle on the ENGINE's ReLU decisions: worst L2 {wf[0]:.3g} ({wf[2]}), max {wfm[1]:.3g} ({wfm[2]}); strict "
                     f"gate 1e-3 / 5e-3: {len(bad_forced)} outside")
        if bad_forced:
            failures.append(("f32 gradient tensors outside the strict gate (engine's ReLU decisions)", bad_forced[:6]))
        far = [t for t in outside if t[1] > 3e-3 or t[2] > 1.5e-2]
        if far:
            failures.append(("f32 gradient tensors beyond 3e-3 / 1.5e-2 of the float64 oracle", far[:6]))
    del net, stp, g32
    torch.cuda.empty_cache()

    # ---- bf16 engine (the benchmarked kernels) vs the fp32 oracle, then vs the bf16-emulating oracle -----------------------------
    net, stp = _engine_run("bf16", B, F, tup, deterministic=False)
    eg = stp.grads.cpu()
    G, R = _engine_flat(net, eg, g64), _flat(net, g64)
    c_all, nr_all = (torch.dot(G, R) / (G.norm() * R.norm())).item(), (G.norm() / R.norm()).item()
    bad = []
    for nm, r in g64.items():
        s = net.slots[nm]
        g = eg[s.offset:s.offset + s.numel].double()
        c, ratio = cosine(g, r), (g.norm() / r.norm().clamp_min(1e-300)).item()
        small = r.dim() == 1
        if c < (0.95 if small else 0.97) or abs(ratio - 1.0) > (0.20 if small else 0.12):
            bad.append((nm, c, ratio))
    qerr16 = ((stp.q_before.double().cpu() - q64).abs().max() / q64.abs().max()).item()
    notes.append(f"bf16 engine vs fp32/float64 oracle: loss {stp.loss.item():.6f} / {loss64:.6f}, Q(s) max rel err {qerr16:.2e}, whole-gradient cosine "
                 f"{c_all:.5f}, norm ratio {nr_all:.4f}, {len(bad)} tensors outside the per-tensor cosine / norm gate")
    if c_all < 0.999 or abs(nr_all - 1.0) > 0.02:
        failures.append(("bf16 whole gradient vs oracle", c_all, nr_all))
    if bad:
        failures.append(("bf16 per-tensor vs oracle", bad[:6]))
    if qerr16 > 4e-2 or abs(stp.loss.item() - loss64) > 5e-2 * abs(loss64):
        failures.append(("bf16 loss / Q", stp.loss.item(), loss64, qerr16))
    _bf16_emulation_check(B, F, tup, net, stp, failures, notes)
    warnings.warn(f"full-size oracle parity B={B} F={F} ({time.time() - t0:.0f} s): " + " || ".join(notes))
    assert not failures, failures


@pytest.mark.parametrize("B,F,seed", [(8, 1, 101), (3, 4, 301)], ids=["C1_batch8", "panorama_batch3"])
def test_bf16_step_vs_bf16_emulating_oracle_small(B, F, seed):
    """The same element-wise, in-step check of the bf16 kernels at the small geometries of tests/test_gpu_engine.py (the
    minibatches of its f32 gates): stored activations at 2e-2, every gradient tensor at 1.5e-2 relative L2 against the oracle
    that rounds where the engine rounds and takes its ReLU masks (measured worst 1.12e-2, `_bf16_emulation_check`)."""
    (tup, _) = synth.make_batch(seed, B, F, structured=True, reward_p=0.3)
    net, stp = _engine_run("bf16", B, F, tup, deterministic=False)
    failures, notes = [], []
    _bf16_emulation_check(B, F, tup, net, stp, failures, notes)
    warnings.warn(f"bf16 emulation parity B={B} F={F}: " + " || ".join(notes))
    assert not failures, failures
