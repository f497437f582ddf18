"""Decoded-frame shards == the JPEG loader, sample by sample (frames bit-exact, labels equal), incl. the
PREVIOUS_IMAGES 4-frame gather clamped at the episode start and the VALUE_LEARNING targets."""
import numpy as np
import torch


def _make_dataset(tmp_path, n=9):
    import pandas as pd
    from PIL import Image
    from video_dqn_amd.dataset import detection_thresholds
    rng = np.random.default_rng(3)
    ep = tmp_path / "frames" / "ep0"
    ep.mkdir(parents=True)
    for i in range(1, n + 4):
        Image.fromarray(rng.integers(0, 256, (250 + i, 330, 3), dtype=np.uint8)).save(ep / ("%04d.jpg" % i), quality=92)
    rows = []
    for i in range(2, n + 1):  # the episode starts at frame 2: frames id-1.. clamp there
        row = {"before_image": str(ep / ("%04d.jpg" % i)), "after_image": str(ep / ("%04d.jpg" % (i + 3))), "ep_id": 0,
               "im_start": 2, "im_stop": n + 4, "inverse_actions": int(i % 3)}
        for c in range(5):
            row[f"detector_score{c}"] = float(detection_thresholds[c] + (0.01 if (i + c) % 4 == 0 else -0.02))
            row[f"sparse_reward{c}"] = int((i + c) % 4 == 0)
            row[f"steps_to_reward{c}"] = float((i + c) % 4) if c != 2 else np.inf
            row[f"steps_to_reward_neg{c}"] = 0.0
        rows.append(row)
    pd.DataFrame(rows).to_feather(tmp_path / "data.feather")
    return str(tmp_path / "data.feather")


def test_shards_equal_jpeg_loader(tmp_path):
    from video_dqn_amd.dataset import QLearningRealDataset
    from video_dqn_amd.shards import ShardDataset, build_shards, is_shard_dir
    feather = _make_dataset(tmp_path)
    out = str(tmp_path / "shards")
    info = build_shards(feather, out, shard_frames=4, log=lambda *a: None)
    assert is_shard_dir(out) and info["samples"] == 8 and info["shards"] == (info["frames"] + 3) // 4
    assert info["frames"] <= 12  # consecutive samples share frames: far fewer decodes than 2 (or 8) per sample
    for kw in (dict(inverse_actions=True), dict(one_action=True, value_learning=True),
               dict(inverse_actions=True, previous_images=True), dict(one_action=True, confidence_reward=True)):
        a = QLearningRealDataset(feather, as_uint8=True, **kw)
        b = ShardDataset(out, **kw)
        assert len(a) == len(b) and abs(a.reward_percentage() - b.reward_percentage()) < 1e-12
        for i in range(len(a)):
            ta, tb = a[i], b[i]
            assert torch.equal(ta[0], tb[0]) and torch.equal(ta[1], tb[1])
            assert ta[2] == tb[2]
            np.testing.assert_array_equal(ta[3], tb[3])
            np.testing.assert_array_equal(ta[5], tb[5])
            np.testing.assert_array_equal(ta[6], tb[6])
    batch = next(iter(torch.utils.data.DataLoader(ShardDataset(out, inverse_actions=True, previous_images=True), batch_size=4)))
    assert batch[0].shape == (4, 4, 224, 224, 3) and batch[0].dtype == torch.uint8
