"""GPU tests of the drop-in boundary: the product HabitatDQNMultiAction module, run_train with checkpoints,
resume, load_model_number (what evaluation/runner.py:61 calls)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import relerr  # noqa: E402
from video_dqn_amd import synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_module_forward_eval_and_b1_quirk():
    from oracle import ref_cpu
    from video_dqn_amd.model import HabitatDQNMultiAction
    m = HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False, dtype="f32", device="cuda")
    sd = synth.make_state_dict(11)
    m.load_state_dict(sd, strict=True)
    m.eval()
    ref = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False)
    ref.load_state_dict(sd)
    ref.eval()
    (tup, raw) = synth.make_batch(24, 3, 1, structured=True)
    with torch.no_grad():
        q = m(tup[0].cuda())
        r = ref(tup[0])
    assert q.shape == (3, 5, 3) and q.dtype == torch.float32
    assert relerr(q, r) < 1e-3
    # CPU input tensors are moved like the reference's caller would have done; B = 1 keeps the (1, 5, A) shape
    q1 = m(tup[0][:1])
    assert q1.shape == (1, 5, 3) and relerr(q1, r[:1]) < 1e-3
    # uint8 frames (util/torch.py:26-36 to_imgnet semantics fused on the GPU), as evaluate.py:110-114 scores views
    qs = m(torch.from_numpy(raw[0][:, 0]).cuda())
    assert relerr(qs, r) < 1e-3
    assert float(qs[0, 2, :].max()) == pytest.approx(float(r[0, 2, :].max()), rel=1e-3)
    # state_dict round trip returns what was loaded
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    with pytest.raises(Exception, match="bad shape"):
        m(torch.zeros(2, 4, 3, 224, 224))


def test_run_train_checkpoint_resume_and_load_model_number(tmp_path):
    from oracle import ref_cpu
    from video_dqn_amd.config import ExperimentConfig
    from video_dqn_amd.model import load_model_number
    from video_dqn_amd.trainer import run_train
    folder = tmp_path / "exp"
    folder.mkdir()
    (folder / "config.yml").write_text(
        "DATASET: 'synthetic'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nLEARNING_RATE: 0.0001\n"
        "GAMMA: 0.99\nCHECKPOINT_INTERVAL: 3\nNUM_STEPS: 6\nTARGET_UPDATE_INTERVAL: 2\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\n"
        "COMPUTE_DTYPE: 'f32'\n")
    cfg = ExperimentConfig(str(folder), device="cuda")
    model, stepper, running = run_train(cfg, max_steps=3)
    assert os.path.exists(folder / "models" / "sample3.torch") and running is not None and np.isfinite(running)
    snap = torch.load(folder / "models" / "sample3.torch", map_location="cpu")
    assert snap["sample_number"] == 3 and len(snap["model_state_dict"]) == 250
    assert sorted(snap["optimizer_state_dict"]["state"].keys()) == [i for i in range(70) if i not in (60, 61)]
    assert snap["optimizer_state_dict"]["state"][0]["step"] == 3
    # the checkpoint loads strictly into the (oracle restatement of the) reference class
    ref = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False)
    ref.load_state_dict(snap["model_state_dict"], strict=True)
    # resume: -r picks sample3, restores model + Adam and continues to NUM_STEPS = 6
    cfg2 = ExperimentConfig(str(folder), device="cuda", resume=True)
    model2, stepper2, _ = run_train(cfg2, resume_from=3)
    # reference quirk kept: sample_number restarts at resume_from + 1 and is incremented before the first update
    # (train_q_network.py:190,213), so a resumed run performs updates 5 and 6 -> 3 + 2 Adam steps
    assert stepper2.adam_step == 5 and os.path.exists(folder / "models" / "sample6.torch")
    # evaluation-side loader (train_q_network.load_model_number, evaluation/runner.py:61)
    cfg3 = ExperimentConfig(str(folder), device="cuda", tensorboard=False)
    m3 = load_model_number(cfg3, 6)
    m3.eval()
    (tup, _) = synth.make_batch(3, 2, 1, structured=True)
    q3 = m3(tup[0].cuda())
    q2 = model2(tup[0].cuda())
    assert torch.equal(q3, q2)


def test_bootstrap_branch_loads_model_and_optimizer(tmp_path):
    """BOOTSTRAP (train_q_network.py:200-206): the run starts from another run's checkpoint — model AND Adam state —
    instead of silently training from scratch; a missing file raises as in the reference."""
    from video_dqn_amd.config import ExperimentConfig
    from video_dqn_amd.trainer import run_train
    base = ("DATASET: 'synthetic'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nLEARNING_RATE: 0.0001\n"
            "GAMMA: 0.99\nCHECKPOINT_INTERVAL: 2\nNUM_STEPS: 2\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\nCOMPUTE_DTYPE: 'f32'\n")
    src = tmp_path / "gt"
    src.mkdir()
    (src / "config.yml").write_text(base)
    m0, st0, _ = run_train(ExperimentConfig(str(src), device="cuda", tensorboard=False))
    ckpt = src / "models" / "sample2.torch"
    dst = tmp_path / "boot"
    dst.mkdir()
    (dst / "config.yml").write_text(base.replace("NUM_STEPS: 2", "NUM_STEPS: 1") + f"BOOTSTRAP: True\nBOOTSTRAP_CHECKPOINT: '{ckpt}'\n")
    logs = []
    m1, st1, _ = run_train(ExperimentConfig(str(dst), device="cuda", tensorboard=False), log=lambda *a: logs.append(" ".join(map(str, a))))
    assert any("BOOTSTRAP" in l for l in logs) and any(str(ckpt) in l for l in logs)
    assert st1.adam_step == 3  # two steps restored from the checkpoint + one update here
    snap = torch.load(ckpt, map_location="cpu")
    w0, w1 = snap["model_state_dict"]["top.4.weight"], m1.state_dict()["top.4.weight"].cpu()
    assert (w0 - w1).abs().max().item() < 3e-4 and not torch.equal(w0, w1)  # one lr=1e-4 Adam step away from the checkpoint
    (dst / "config.yml").write_text(base + "BOOTSTRAP: True\nBOOTSTRAP_CHECKPOINT: '/nonexistent/epoch99.torch'\n")
    with pytest.raises(FileNotFoundError):
        run_train(ExperimentConfig(str(dst), device="cuda", tensorboard=False))


def test_run_train_basic_arch_defaults(tmp_path):
    """defaults.py's own configuration: ARCHITECTURE 'basic' + PANORAMA (F = 4): the loop trains, BatchNorm statistics and
    num_batches_tracked advance, and the checkpoint loads strictly into the reference class layout (244 keys)."""
    from oracle import ref_cpu
    from video_dqn_amd.config import ExperimentConfig
    from video_dqn_amd.trainer import run_train
    folder = tmp_path / "exp"
    folder.mkdir()
    (folder / "config.yml").write_text(
        "DATASET: 'synthetic'\nCHECKPOINT_INTERVAL: 2\nNUM_STEPS: 2\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\nCOMPUTE_DTYPE: 'bf16'\n")
    cfg = ExperimentConfig(str(folder), device="cuda")
    assert cfg.ARCHITECTURE == "basic" and cfg.PANORAMA is True
    model, stepper, running = run_train(cfg)
    assert running is not None and np.isfinite(running)
    snap = torch.load(folder / "models" / "sample2.torch", map_location="cpu")
    sd = snap["model_state_dict"]
    assert len(sd) == 244 and int(sd["resnet.bn1.num_batches_tracked"]) == 2 * 2 * 4
    assert int(sd["resnet.layer4.1.bn2.num_batches_tracked"]) == 16
    assert not torch.equal(sd["resnet.bn1.running_mean"], torch.zeros(64))
    ref = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=False, panorama=True)
    ref.load_state_dict(sd, strict=True)
    assert sorted(snap["optimizer_state_dict"]["state"].keys()) == [i for i in range(64) if i not in (60, 61)]
    # eval-mode forward of the trained model agrees with the oracle evaluated from the checkpoint
    (tup, _) = synth.make_batch(3, 2, 4, structured=True)
    model.eval()
    ref.eval()
    with torch.no_grad():
        from helpers import relerr
        assert relerr(model(tup[0].cuda()), ref(tup[0])) < 6e-2


def test_run_train_on_feather_jpeg_dataset_and_shards(tmp_path):
    """C1-style plumbing on real files: config dir -> feather + JPEG frames -> loader (uint8, normalise fused on the GPU)
    -> TD updates -> checkpoint; the same run from decoded-frame shards sees identical batches, so with one loader worker
    and the same seed the two runs end with the same parameters; the inverse-action labelling utility writes the column
    the loader reads."""
    import PIL  # noqa: F401  (hard requirements of this data path: a missing one must fail the run, not skip it)
    import pyarrow  # noqa: F401
    from test_shards_cpu import _make_dataset
    from video_dqn_amd.config import ExperimentConfig
    from video_dqn_amd.shards import build_shards
    from video_dqn_amd.trainer import run_train
    feather = _make_dataset(tmp_path, n=9)
    shards = str(tmp_path / "shards")
    build_shards(feather, shards, shard_frames=4, log=lambda *a: None)
    finals = []
    for tag, dataset in (("jpeg", feather), ("shards", shards)):
        folder = tmp_path / f"exp_{tag}"
        folder.mkdir()
        (folder / "config.yml").write_text(
            f"DATASET: '{dataset}'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nLEARNING_RATE: 0.0001\n"
            "GAMMA: 0.99\nUSE_INVERSE_ACTIONS: True\nCHECKPOINT_INTERVAL: 2\nNUM_STEPS: 2\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\n"
            "COMPUTE_DTYPE: 'f32'\n")
        cfg = ExperimentConfig(str(folder), device="cuda", tensorboard=False)
        model, stepper, running = run_train(cfg)
        assert os.path.exists(folder / "models" / "sample2.torch") and np.isfinite(running)
        finals.append(model.engine.params.cpu().clone())
    assert torch.equal(finals[0], finals[1]) or (finals[0] - finals[1]).abs().max().item() < 5e-4  # f32 atomics order only

    # frames resident in HBM: device-side gather == the loader's batches for the same indices; the trainer runs from it
    from video_dqn_amd.shards import DeviceFrameStore, ShardDataset
    for kw in (dict(inverse_actions=True), dict(one_action=True, value_learning=True, previous_images=True)):
        store = DeviceFrameStore(shards, "cuda", **kw)
        ds = ShardDataset(shards, **kw)
        ds.batched_fetch = True
        idx = [5, 0, 7, 3]
        ref = ds.__getitems__(idx)[0]
        got = store.gather(torch.tensor(idx, device="cuda"))
        assert torch.equal(got[0].cpu(), ref[0]) and torch.equal(got[1].cpu(), ref[1]) and got[2] == 0
        assert torch.equal(got[3].cpu(), ref[2]) and torch.equal(got[4].cpu(), ref[3].float()) and torch.equal(got[5].cpu(), ref[4].float())
        assert torch.equal(got[6].cpu(), ref[6].float())
        gt_ref = ref[5].float() if ref[5].dim() == 2 else ref[5].float().view(-1, 1).expand(-1, 5)
        assert torch.equal(torch.nan_to_num(got[7].cpu(), nan=-1.0), torch.nan_to_num(gt_ref, nan=-1.0))
    folder = tmp_path / "exp_resident"
    folder.mkdir()
    (folder / "config.yml").write_text(
        f"DATASET: '{shards}'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nUSE_INVERSE_ACTIONS: True\n"
        "CHECKPOINT_INTERVAL: 3\nNUM_STEPS: 3\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\nDEVICE_RESIDENT_DATA: 'on'\n")
    logs = []
    cfg = ExperimentConfig(str(folder), device="cuda", tensorboard=False)
    model, stepper, running = run_train(cfg, log=lambda *a: logs.append(" ".join(str(x) for x in a)))
    assert any("resident in HBM" in l for l in logs) and np.isfinite(running)
    assert os.path.exists(folder / "models" / "sample3.torch")


def test_host_frame_stream_device_branch(tmp_path):
    """The streaming input path's DEVICE branch (replaces dataloaders/q_learning_real.py:55-73 under torch's
    DataLoader, train_q_network.py:98,114): pinned staging slots reused behind an event, host-to-device copies on a prefetch
    stream, record_stream towards the consumer (shards.py HostFrameStream._produce / batches).  Against DeviceFrameStore.batches
    bit for bit — frames, labels, order — over 5 x depth minibatches and two epoch boundaries, with and without the
    PREVIOUS_IMAGES gather, while the COMPUTE stream is kept busy so that every batch is read late: the producer runs ahead of
    the consumer by the queue's depth, re-fills pinned slots and gets recycled device blocks from the allocator — a slot reused
    before its copy finished, or a device buffer recycled before the consumer read it, shows as a wrong frame here."""
    from test_shards_cpu import _synthetic_shards
    from video_dqn_amd.shards import DeviceFrameStore, HostFrameStream
    root = str(tmp_path / "shards")
    _synthetic_shards(root)
    depth, B = 2, 4
    for kw in (dict(inverse_actions=True), dict(inverse_actions=True, previous_images=True)):
        store = DeviceFrameStore(root, "cuda", **kw)
        ref = store.batches(B, 11)
        with HostFrameStream(root, "cuda", B, 11, threads=3, depth=depth, **kw) as stream:
            got = stream.batches()
            held = []
            for k in range(5 * depth + 8):  # 37 // 4 = 9 batches per epoch: 18 batches cross two epoch boundaries
                b = next(got)
                torch.cuda._sleep(20_000_000)  # ~10 ms of compute-stream work queued in front of this batch's first reader
                held.append(tuple(t.clone() if torch.is_tensor(t) else t for t in b))  # the late read, on the compute stream
                del b
            torch.cuda.synchronize()
            for h in held:
                a = next(ref)
                assert a[2] == h[2] == 0
                for x, y in zip(a[:2] + a[3:], h[:2] + h[3:]):
                    assert x.dtype == y.dtype and x.shape == y.shape and y.is_cuda
                    assert torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0))
            thread = stream._thread
        assert stream._thread is None and not thread.is_alive() and stream._slots == []  # close() joined the producer


def test_run_train_streaming_equals_resident(tmp_path):
    """run_train on decoded-frame shards with DEVICE_RESIDENT_DATA 'off' + SHARD_INPUT 'stream' (HostFrameStream) ends with the
    parameters of the HBM-resident run: same minibatch sequence, same updates (deterministic sums: bit for bit)."""
    from test_shards_cpu import _make_dataset
    from video_dqn_amd.config import ExperimentConfig
    from video_dqn_amd.shards import build_shards
    from video_dqn_amd.trainer import run_train
    feather = _make_dataset(tmp_path, n=9)
    shards = str(tmp_path / "shards")
    build_shards(feather, shards, shard_frames=4, log=lambda *a: None)
    finals = []
    for tag, extra in (("resident", "DEVICE_RESIDENT_DATA: 'on'\n"), ("stream", "DEVICE_RESIDENT_DATA: 'off'\nSHARD_INPUT: 'stream'\nHOST_GATHER_THREADS: 2\n")):
        folder = tmp_path / f"exp_{tag}"
        folder.mkdir()
        (folder / "config.yml").write_text(
            f"DATASET: '{shards}'\nPANORAMA: False\nLOSS_CLIP: 'rect'\nARCHITECTURE: 'extra_capacity'\nUSE_INVERSE_ACTIONS: True\n"
            "LEARNING_RATE: 0.0001\nCHECKPOINT_INTERVAL: 5\nNUM_STEPS: 5\nSEED: 4\nBATCH_SIZE: 4\nNUM_WORKERS: 0\nCOMPUTE_DTYPE: 'f32'\n"
            "DETERMINISTIC: True\n" + extra)
        logs = []
        cfg = ExperimentConfig(str(folder), device="cuda", tensorboard=False)
        model, stepper, running = run_train(cfg, log=lambda *a: logs.append(" ".join(str(x) for x in a)))
        assert any(("resident in HBM" if tag == "resident" else "streamed from memory-mapped shards") in l for l in logs), logs
        assert np.isfinite(running) and os.path.exists(folder / "models" / "sample5.torch")
        finals.append((model.engine.params.cpu().clone(), running))
    assert torch.equal(finals[0][0], finals[1][0]) and finals[0][1] == finals[1][1]
