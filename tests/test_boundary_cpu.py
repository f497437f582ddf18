"""CPU-side tests of the drop-in boundary: C-ABI exports, config loader, state_dict layout of the product
model, checkpoint format (loads into the oracle / torch.optim.Adam), dataset semantics, sharding."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """libvdqn.so loads and exports exactly what include/vdqn.h declares (no compute calls here)."""
    from video_dqn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vdqn.h")).read()
    declared = set(re.findall(r"\b(vdqn_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.load().vdqn_abi_version() == _lib.ABI_VERSION


def test_binding_structs_match_the_library():
    """Every argument struct of the ctypes binding has the size the library was compiled with (vdqn_abi_struct_size, checked by
    _lib.load() too): a field added on one side only fails at load time.  The C layout is derived from include/vdqn.h itself:
    the number of fields of each typedef'd struct in the header equals the binding's."""
    from video_dqn_amd import _lib
    lib = _lib.load()
    structs = (("vdqn_conv_args", _lib.ConvArgs), ("vdqn_wgrad_args", _lib.WgradArgs), ("vdqn_td_args", _lib.TdArgs),
               ("vdqn_net_config", _lib.NetConfig), ("vdqn_param_info", _lib.ParamInfo), ("vdqn_prof_entry", _lib.ProfEntry),
               ("vdqn_step_args", _lib.StepArgs))
    hdr = open(os.path.join(ROOT, "include", "vdqn.h")).read()
    for which, (cname, st) in enumerate(structs):
        assert lib.vdqn_abi_struct_size(which) == ctypes.sizeof(st), cname
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_fields = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert n_fields == len(st._fields_), (cname, n_fields, len(st._fields_))
    assert lib.vdqn_abi_struct_size(len(structs)) == -1


def test_compute_fails_loudly_without_gpu():
    from video_dqn_amd import _lib
    from video_dqn_amd.engine import NetEngine, TDStepper
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.VdqnError):
        NetEngine(3, 5, 1, True, "bf16", 8)            # no device given -> needs the GPU
    net = NetEngine(3, 5, 1, True, "bf16", 8, device="cpu")  # storage-only
    with pytest.raises(_lib.VdqnError):
        net.forward(torch.zeros(1, 3, 224, 224), 1, 1)
    with pytest.raises(_lib.VdqnError):
        TDStepper(net, 4, 1e-4, 0.99, True)
    basic = NetEngine(3, 5, 4, False, "bf16", 8, device="cpu")  # ARCHITECTURE='basic': storage-only as well
    with pytest.raises(_lib.VdqnError):
        basic.forward_train(torch.zeros(1, 4, 3, 224, 224), 1, 1)


def test_config_defaults_and_merge(tmp_path):
    from video_dqn_amd.config import ExperimentConfig, get_cfg_defaults
    d = get_cfg_defaults()
    assert d.GAMMA == 0.9 and d.TARGET_UPDATE_INTERVAL == 8000 and d.ARCHITECTURE == "basic" and d.LOSS_CLIP == "none"
    folder = tmp_path / "exp"
    folder.mkdir()
    (folder / "config.yml").write_text(open(os.path.join(ROOT, "configs/experiments/real_data/config.yml")).read())
    c = ExperimentConfig(str(folder), device="cpu", tensorboard=False)
    assert (c.LOSS_CLIP, c.ARCHITECTURE, c.GAMMA, c.LEARNING_RATE, c.NUM_STEPS, c.SEED) == ("rect", "extra_capacity", 0.99, 1e-4, 300000, 4)
    assert c.USE_INVERSE_ACTIONS is True and c.PANORAMA is False and c.CHECKPOINT_INTERVAL == 25000
    assert c.log_dir.endswith("run1")
    (folder / "run1").mkdir()
    (folder / "run3").mkdir()
    assert ExperimentConfig(str(folder), device="cpu", tensorboard=False).log_dir.endswith("run4")
    assert ExperimentConfig(str(folder), device="cpu", tensorboard=False, resume=True).log_dir.endswith("run3")
    (folder / "config.yml").write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        ExperimentConfig(str(folder), device="cpu", tensorboard=False)
    (folder / "config.yml").write_text("LOSS_CLIP: 'bogus'\n")
    with pytest.raises(Exception, match="Invalid value"):
        ExperimentConfig(str(folder), device="cpu", tensorboard=False)
    (folder / "config.yml").write_text("GAMMA: 'high'\n")
    with pytest.raises(ValueError):
        ExperimentConfig(str(folder), device="cpu", tensorboard=False)


def test_model_state_dict_matches_reference_layout(g1):
    """Key set, order, shapes and aliasing of the product model == the reference class (golden G1)."""
    from video_dqn_amd.model import HabitatDQNMultiAction
    for ec, pano in ((True, False), (True, True), (False, False), (False, True)):
        ref = g1[f"ec{int(ec)}_pano{int(pano)}"]
        m = HabitatDQNMultiAction(3, 5, extra_capacity=ec, panorama=pano, device="cpu")
        sd = m.state_dict()
        assert list(sd.keys()) == ref["keys"]
        assert [list(v.shape) for v in sd.values()] == ref["shapes"]
        ptr, alias = {}, []
        for k, v in sd.items():
            alias.append(ptr.setdefault(v.data_ptr(), k) if v.numel() > 0 else k)
        assert alias == ref["alias"]
        assert [n for n, _ in m.named_parameters()] == ref["params"]
    with pytest.raises(Exception, match="bad shape"):
        HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False, device="cpu")(torch.zeros(2, 4, 3, 224, 224))


def test_pretrained_weights_land_in_the_trunk(tmp_path):
    """`models.resnet18(pretrained=True)` (archs/HabitatDQNMultiAction.py:11) served from a file: a torchvision-keyed ResNet-18
    state_dict (the oracle's restated trunk has torchvision's module names) reaches EVERY trunk tensor — 20 convolutions, 20
    BatchNorm layers with running statistics, the frozen fc — bit for bit, the head keeps its own initialisation, wrong shapes
    and partial files are errors that leave the model untouched."""
    from oracle import ref_cpu
    from video_dqn_amd.model import HabitatDQNMultiAction, load_torchvision_resnet18
    torch.manual_seed(3)
    trunk = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False).resnet
    with torch.no_grad():
        for t in trunk.state_dict().values():  # distinct non-default values everywhere (fresh BatchNorm buffers are 0 / 1)
            if t.dtype.is_floating_point:
                t.copy_(torch.rand_like(t) + 0.5)
    sd = trunk.state_dict()
    assert "conv1.weight" in sd and "layer4.1.bn2.running_var" in sd and "fc.bias" in sd and "bn1.num_batches_tracked" in sd
    path = tmp_path / "resnet18.pth"
    torch.save(sd, path)
    for ec in (True, False):
        torch.manual_seed(11)
        base = HabitatDQNMultiAction(3, 5, extra_capacity=ec, panorama=False, device="cpu")
        torch.manual_seed(11)
        m = HabitatDQNMultiAction(3, 5, extra_capacity=ec, panorama=False, device="cpu", pretrained_weights=str(path))
        assert m.pretrained_loaded and not base.pretrained_loaded
        got = m.state_dict()
        n_trunk = 0
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            assert torch.equal(got["resnet." + k], v), k
            n_trunk += 1
        assert n_trunk == 20 + 20 * 4 + 2  # conv weights, BatchNorm (weight, bias, mean, var), fc
        assert n_trunk == sum(1 for n in m.engine.slots if n.startswith("resnet."))
        for k in got:  # the head: untouched by the file (same seeded initialisation as without it)
            if k.startswith("top.") or (ec and k.startswith("features.8.")):
                assert torch.equal(got[k], base.state_dict()[k]), k
    m = HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False, device="cpu")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    partial = {k: v for k, v in sd.items() if not k.startswith("layer4.")}
    with pytest.raises(ValueError, match="trunk tensors missing"):
        load_torchvision_resnet18(m.engine, partial)
    wrong = dict(sd)
    wrong["layer2.0.conv1.weight"] = torch.zeros(128, 64, 1, 1)
    with pytest.raises(ValueError, match="wrong shapes"):
        load_torchvision_resnet18(m.engine, wrong)
    with pytest.raises(ValueError, match="unknown keys"):
        load_torchvision_resnet18(m.engine, dict(sd, **{"layer5.0.conv1.weight": torch.zeros(1)}))
    with pytest.raises(ValueError):
        torch.save({"epoch": 3}, tmp_path / "junk.pth")
        HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False, device="cpu", pretrained_weights=str(tmp_path / "junk.pth"))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    # wrapped forms: {'state_dict': ...} with DataParallel's 'module.' prefix (the Places365 release format)
    assert load_torchvision_resnet18(m.engine, {"state_dict": {"module." + k: v for k, v in sd.items()}}) == 102
    assert torch.equal(m.state_dict()["resnet.layer3.1.conv2.weight"], sd["layer3.1.conv2.weight"])
    # a Places365 ResNet-18 (365-way classifier): every tensor the Q-network computes with is loaded, the off-path fc is skipped
    m2 = HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False, device="cpu")
    fc_before = m2.state_dict()["resnet.fc.weight"].clone()
    places = {"state_dict": {"module." + k: v for k, v in sd.items()}}
    places["state_dict"]["module.fc.weight"], places["state_dict"]["module.fc.bias"] = torch.zeros(365, 512), torch.zeros(365)
    with pytest.warns(UserWarning, match="classifier fc"):
        assert load_torchvision_resnet18(m2.engine, places) == 100
    assert torch.equal(m2.state_dict()["resnet.layer4.1.bn2.running_var"], sd["layer4.1.bn2.running_var"])
    assert torch.equal(m2.state_dict()["resnet.fc.weight"], fc_before)


def test_checkpoint_roundtrip_into_oracle_and_adam(tmp_path):
    """A checkpoint written from the flat engine state loads (strict) into the oracle restatement of the
    reference class and into torch.optim.Adam, and back into the product model bit-exactly."""
    from oracle import ref_cpu
    from video_dqn_amd import synth
    from video_dqn_amd.model import HabitatDQNMultiAction
    from video_dqn_amd.trainer import load_optimizer_state_dict, optimizer_state_dict
    m = HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False, device="cpu")
    sd0 = synth.make_state_dict(5)
    m.load_state_dict(sd0, strict=True)

    class FakeStepper:  # the Adam-state part of TDStepper without a GPU
        pass
    st = FakeStepper()
    st.net, st.lr, st.betas, st.eps, st.adam_step = m.engine, 1e-4, (0.9, 0.999), 1e-8, 3
    nt = m.engine.trainable_numel
    st.exp_avg = torch.from_numpy(synth.uniform(1, "m", (nt,), -1e-3, 1e-3))
    st.exp_avg_sq = torch.from_numpy(synth.uniform(1, "v", (nt,), 0, 1e-6))
    path = tmp_path / "sample3.torch"
    torch.save({"sample_number": 3, "model_state_dict": m.state_dict(), "optimizer_state_dict": optimizer_state_dict(st)}, path)
    snap = torch.load(path, map_location="cpu")
    assert set(snap.keys()) == {"sample_number", "model_state_dict", "optimizer_state_dict"}
    ref = ref_cpu.HabitatDQNMultiAction(3, 5, extra_capacity=True, panorama=False)
    ref.load_state_dict(snap["model_state_dict"], strict=True)
    for k, v in ref.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    opt = torch.optim.Adam(ref.parameters(), lr=1.0)
    opt.load_state_dict(snap["optimizer_state_dict"])
    assert opt.param_groups[0]["lr"] == 1e-4 and len(opt.state_dict()["state"]) == 68
    assert 60 not in opt.state_dict()["state"] and 61 not in opt.state_dict()["state"]
    # Adam can step from it (the state is complete and well-formed)
    for p in ref.parameters():
        p.grad = torch.zeros_like(p) if p.shape != ref.resnet.fc.weight.shape and p.shape != ref.resnet.fc.bias.shape else None
    opt.step()
    # and back (re-read: torch's Adam aliases the loaded state tensors and has just stepped them in place)
    snap = torch.load(path, map_location="cpu")
    st2 = FakeStepper()
    st2.net, st2.exp_avg, st2.exp_avg_sq = m.engine, torch.zeros(nt), torch.zeros(nt)
    load_optimizer_state_dict(st2, snap["optimizer_state_dict"])
    assert st2.adam_step == 3 and st2.lr == 1e-4
    pad = torch.ones(nt, dtype=torch.bool)
    for s in m.engine.slots.values():
        if s.kind == 0:
            pad[s.offset:s.offset + s.numel] = False
    assert torch.equal(st2.exp_avg[~pad], st.exp_avg[~pad]) and torch.equal(st2.exp_avg_sq[~pad], st.exp_avg_sq[~pad])


def test_dataset_semantics(tmp_path):
    """feather row -> 7-tuple exactly as dataloaders/q_learning_real.py:55-98 builds it."""
    import pandas as pd
    from PIL import Image
    from video_dqn_amd.dataset import QLearningRealDataset, detection_thresholds, image_net_transform
    rng = np.random.default_rng(0)
    rows = []
    for i in range(6):
        for tag in ("b", "a"):
            Image.fromarray(rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)).save(tmp_path / f"{tag}{i:04d}.jpg")
        row = {"before_image": str(tmp_path / f"b{i:04d}.jpg"), "after_image": str(tmp_path / f"a{i:04d}.jpg"),
               "ep_id": 0, "im_start": 0, "im_stop": 5, "inverse_actions": i % 3}
        for c in range(5):
            row[f"detector_score{c}"] = float(detection_thresholds[c] + (0.01 if (i + c) % 3 == 0 else -0.01))
            row[f"sparse_reward{c}"] = int((i + c) % 3 == 0)
            row[f"steps_to_reward{c}"] = float(c) if (i + c) % 2 == 0 else np.inf
            row[f"steps_to_reward_neg{c}"] = 0.0
        rows.append(row)
    df = pd.DataFrame(rows)
    df.to_feather(tmp_path / "data.feather")
    ds = QLearningRealDataset(str(tmp_path / "data.feather"), one_action=True, inverse_actions=True)
    bi, ai, action, reward, term, gt, vm = ds[2]
    assert bi.shape == (3, 224, 224) and bi.dtype == torch.float32 and ai.shape == (3, 224, 224)
    assert action == 2 and list(reward) == [int((2 + c) % 3 == 0) for c in range(5)] and reward is term
    assert np.isnan(gt) and list(vm) == [1] * 5 and reward.dtype == np.int64
    assert abs(ds.reward_percentage() - np.mean([max((i + c) % 3 == 0 for c in range(5)) for i in range(6)])) < 1e-12
    du = QLearningRealDataset(str(tmp_path / "data.feather"), one_action=True, as_uint8=True)
    u8 = du[2][0]
    assert u8.dtype == torch.uint8 and u8.shape == (224, 224, 3) and du[2][2] == 0
    assert torch.equal(image_net_transform(u8.numpy()), bi)      # same pixels, normalisation is the only difference
    dv = QLearningRealDataset(str(tmp_path / "data.feather"), one_action=True, value_learning=True)
    g = dv[1][5]
    assert g.shape == (5,) and np.isnan(g[0]) and abs(g[1] - 0.99 ** 1) < 1e-12
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, drop_last=True)))
    assert batch[0].shape == (4, 3, 224, 224) and batch[2].dtype == torch.int64 and batch[3].shape == (4, 5) and batch[5].dtype == torch.float64


def test_shard_indices_disjoint_and_equal():
    from video_dqn_amd.dist import shard_indices
    perm = list(np.random.default_rng(1).permutation(103))
    shards = [shard_indices(103, r, 4, 8, perm) for r in range(4)]
    assert all(len(s) == 24 for s in shards)
    flat = [i for s in shards for i in s]
    assert len(set(flat)) == len(flat)


@pytest.mark.refonly
@pytest.mark.skipif(not os.path.isdir("/root/reference/archs"), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("extra_capacity,panorama,n_keys,n_ids", [(True, False, 250, 70), (False, True, 244, 64)])
def test_checkpoint_loads_into_the_reference_class(tmp_path, extra_capacity, panorama, n_keys, n_ids):
    """SURVEY 8c G6 as written: a checkpoint in the build's format (trainer.save_checkpoint — what run_train writes every
    CHECKPOINT_INTERVAL, train_q_network.py:241-247) -> the REFERENCE's own class, imported from /root/reference exactly as
    tests/golden/make_golden.py does (torchvision is not installed here: `models.resnet18` is the published topology restated in
    oracle/ref_cpu.py), through the reference's own loading lines: `model.load_state_dict(snapshot['model_state_dict'])`
    (train_q_network.py:50-57, strict) and `optimizer.load_state_dict(snapshot['optimizer_state_dict'])` (:197); the reference model
    then returns the tensors the build saved, and its Adam steps from the loaded state."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    RefClass = mg.import_reference_model()
    from video_dqn_amd import synth
    from video_dqn_amd.model import HabitatDQNMultiAction
    from video_dqn_amd.trainer import optimizer_state_dict, save_checkpoint
    m = HabitatDQNMultiAction(3, 5, extra_capacity=extra_capacity, panorama=panorama, device="cpu")
    if extra_capacity:
        m.load_state_dict(synth.make_state_dict(5), strict=True)

    class FakeStepper:  # the Adam-state part of TDStepper without a GPU
        pass
    st = FakeStepper()
    st.net, st.lr, st.betas, st.eps, st.adam_step = m.engine, 1e-4, (0.9, 0.999), 1e-8, 3
    nt = m.engine.trainable_numel
    st.exp_avg = torch.from_numpy(synth.uniform(1, "m", (nt,), -1e-3, 1e-3))
    st.exp_avg_sq = torch.from_numpy(synth.uniform(1, "v", (nt,), 0, 1e-6))
    path = tmp_path / "sample3.torch"
    save_checkpoint(path, 3, m, st)
    snapshot = torch.load(path, map_location="cpu")
    assert len(snapshot["model_state_dict"]) == n_keys
    ref = RefClass(3, 5, extra_capacity=extra_capacity, panorama=panorama)
    assert type(ref).__module__ == "archs.HabitatDQNMultiAction"
    ref.load_state_dict(snapshot["model_state_dict"])  # train_q_network.py:56 (strict is the default)
    saved = m.state_dict()
    for k, v in ref.state_dict().items():
        assert torch.equal(v, saved[k]), k
    optimizer = torch.optim.Adam(ref.parameters(), lr=0.5)  # :124 builds it over model.parameters(); the loaded group carries the lr
    optimizer.load_state_dict(snapshot["optimizer_state_dict"])  # :197
    assert optimizer.param_groups[0]["lr"] == 1e-4 and len(optimizer.param_groups[0]["params"]) == n_ids
    frozen = {id(p) for p in (ref.resnet.fc.weight, ref.resnet.fc.bias)}
    assert all(id(p) not in optimizer.state for p in ref.parameters() if id(p) in frozen)
    for p in ref.parameters():
        p.grad = None if id(p) in frozen else torch.zeros_like(p)
    optimizer.step()
    assert all(int(s["step"]) == 4 for s in optimizer.state.values())
