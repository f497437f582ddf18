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


def _synthetic_shards(root, n_frames=24, shard_frames=8, n_samples=37):
    import os
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(5)
    for s in range(n_frames // shard_frames):
        mm = np.lib.format.open_memmap(os.path.join(root, f"frames_{s:05d}.npy"), mode="w+", dtype=np.uint8, shape=(shard_frames, 224, 224, 3))
        mm[:] = rng.integers(0, 256, (shard_frames, 224, 224, 3), dtype=np.uint8)
        mm.flush()
    np.savez(os.path.join(root, "index.npz"), before=rng.integers(0, n_frames, (n_samples, 4)), after=rng.integers(0, n_frames, (n_samples, 4)),
             shard_frames=np.int64(shard_frames), n_frames=np.int64(n_frames), detector_score=rng.random((n_samples, 5)),
             sparse_reward=rng.integers(0, 2, (n_samples, 5)), steps_to_reward=rng.random((n_samples, 5)),
             inverse_actions=rng.integers(0, 3, n_samples), has_inverse_actions=np.int64(1), with_previous=np.int64(1))


def test_host_frame_stream_equals_resident_store(tmp_path):
    """The streaming input path (memory-mapped shards -> vdqn_host_gather -> staging buffers -> device) delivers the SAME minibatch
    sequence as the HBM-resident store for the same seed, bit for bit: frames, labels, order — over more than one epoch (the
    per-epoch permutation), with and without the PREVIOUS_IMAGES gather, and for both ranks of a two-rank job (rank-strided slices
    of one permutation).  Runs on the CPU: the gather is host code of libvdqn.so (no GPU call), the device is 'cpu'."""
    from video_dqn_amd.shards import DeviceFrameStore, HostFrameStream
    root = str(tmp_path / "shards")
    _synthetic_shards(root)
    for kw in (dict(inverse_actions=True), dict(inverse_actions=True, previous_images=True), dict(one_action=True, value_learning=True)):
        for world in (1, 2):
            for rank in range(world):
                store = DeviceFrameStore(root, "cpu", **kw)
                ref = store.batches(4, 11, rank, world)
                stream = HostFrameStream(root, "cpu", 4, 11, rank, world, threads=3, depth=2, **kw)
                got = stream.batches()
                per_epoch = (37 // world // 4)
                for _ in range(2 * per_epoch + 3):  # crosses two epoch boundaries
                    a, b = next(ref), next(got)
                    assert a[2] == b[2] == 0
                    for x, y in zip(a[:2] + a[3:], b[:2] + b[3:]):
                        assert x.dtype == y.dtype and x.shape == y.shape
                        assert torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0))
                stream.close()


def test_rank_sharded_store_holds_a_quarter_and_draws_the_same_batches(tmp_path):
    """Four ranks with resident data (round-4 review, 7b): each rank's store holds only the frames its samples of the epoch reference
    — a quarter of the dataset when samples do not share frames — and yields bit for bit the minibatch sequence of the full
    per-rank copy (DeviceFrameStore.batches(B, seed, rank, 4)), across an epoch boundary (the subset is re-uploaded per epoch)."""
    import os
    from video_dqn_amd.shards import FRAME_BYTES, DeviceFrameStore, RankShardedFrameStore
    root = str(tmp_path / "shards")
    n_samples, shard_frames = 32, 16
    os.makedirs(root)
    rng = np.random.default_rng(9)
    n_frames = 2 * n_samples  # every sample has its own before / after frame
    for s in range(n_frames // shard_frames):
        mm = np.lib.format.open_memmap(os.path.join(root, f"frames_{s:05d}.npy"), mode="w+", dtype=np.uint8, shape=(shard_frames, 224, 224, 3))
        mm[:] = rng.integers(0, 256, (shard_frames, 224, 224, 3), dtype=np.uint8)
        mm.flush()
    frames = rng.permutation(n_frames).reshape(n_samples, 2)
    np.savez(os.path.join(root, "index.npz"), before=np.repeat(frames[:, :1], 4, axis=1), after=np.repeat(frames[:, 1:], 4, axis=1),
             shard_frames=np.int64(shard_frames), n_frames=np.int64(n_frames), detector_score=rng.random((n_samples, 5)),
             sparse_reward=rng.integers(0, 2, (n_samples, 5)), steps_to_reward=rng.random((n_samples, 5)),
             inverse_actions=rng.integers(0, 3, n_samples), has_inverse_actions=np.int64(1), with_previous=np.int64(1))
    world, B = 4, 2
    full = DeviceFrameStore(root, "cpu", inverse_actions=True)
    for rank in range(world):
        ref = full.batches(B, 5, rank, world)
        st = RankShardedFrameStore(root, "cpu", rank, world, threads=2, chunk_frames=5, inverse_actions=True)
        got = st.batches(B, 5)
        for k in range(2 * (n_samples // world // B) + 2):
            a, b = next(ref), next(got)
            for x, y in zip(a[:2] + a[3:], b[:2] + b[3:]):
                assert x.shape == y.shape and torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0))
            assert st.bytes() == full.bytes() // world == (n_frames // world) * FRAME_BYTES


def test_host_gather_copies_every_record_once():
    """vdqn_host_gather (csrc/hostio.hip; the per-sample fetch + collate of dataloaders/q_learning_real.py:55-73 for decoded-frame
    shards): n records from n addresses into one contiguous buffer, on 1 .. 9 threads, with n below, equal to and far above the thread
    count, n = 0, and an error (not a crash) for a null destination."""
    import ctypes as C
    import numpy as np
    from video_dqn_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=(37, 1000), dtype=np.uint8)
    for n in (0, 1, 3, 8, 37):
        for threads in (0, 1, 2, 4, 9):
            order = rng.permutation(37)[:n]
            ptrs = (C.c_void_p * max(n, 1))(*[src[i].ctypes.data for i in order])
            dst = np.full((max(n, 1), 1000), 7, dtype=np.uint8)
            rc = lib.vdqn_host_gather(C.c_void_p(dst.ctypes.data), ptrs, n, 1000, threads)
            assert rc == 0
            assert np.array_equal(dst[:n], src[order]) and (n > 0 or (dst == 7).all())
    assert lib.vdqn_host_gather(None, ptrs, 1, 1000, 1) != 0
    assert b"vdqn_host_gather" in lib.vdqn_last_error()
