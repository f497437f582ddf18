#!/usr/bin/env python
"""CLI-compatible entry point: ``python train_q_network.py <config_dir> [-g GPUS] [-r] [-d]``
(reference ``train_q_network.py:253-296``).  The public names evaluation code imports from the reference
module — ``build_model``, ``load_model_number``, ``run_train`` — are re-exported here.

Multi-GPU (new): ``-g 0,1,2,3`` starts one rank per listed GPU from this process (before anything touches the GPU:
video_dqn_amd/launch.py), or launch with ``python -m torch.distributed.run --nproc-per-node N train_q_network.py <dir>``;
each rank takes GPU LOCAL_RANK and a disjoint shard of every shuffled epoch.
"""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime initialises: see video_dqn_amd/__init__.py

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _lazy():
    from video_dqn_amd.model import build_model, load_model_number  # noqa: F401
    from video_dqn_amd.trainer import loopLoader, run_train  # noqa: F401
    return build_model, load_model_number, loopLoader, run_train


def __getattr__(name):  # `from train_q_network import load_model_number` (evaluation/runner.py:10)
    if name in ("build_model", "load_model_number", "loopLoader", "run_train"):
        return dict(zip(("build_model", "load_model_number", "loopLoader", "run_train"), _lazy()))[name]
    raise AttributeError(name)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="train q network")
    parser.add_argument("-g", "--gpu", dest="gpu", default="0", help="which gpu to run on")
    parser.add_argument("-r", "--resume", dest="resume", action="store_true", help="resume from last epoch?")
    parser.add_argument("-d", "--delete", dest="delete", action="store_true", help="delete stored tensorboard data")
    parser.add_argument("--max-steps", type=int, default=None, help="stop after this many updates (testing)")
    parser.add_argument("config", help="folder containing config file")
    args = parser.parse_args()

    from video_dqn_amd import launch
    gpu_ids = [g for g in args.gpu.split(",") if g != ""]
    launch.die_with_parent()  # a rank started by spawn_ranks goes down with its launcher
    if len(gpu_ids) > 1 and not launch.in_rank_env():
        # one rank per listed GPU; this parent never initialises HIP and exits with the ranks' code
        single = os.environ.get("VDQN_SINGLE_DEVICE") == "1"  # functional test: every rank on the first listed GPU
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], len(gpu_ids),
                                    {"HIP_VISIBLE_DEVICES": gpu_ids[0] if single else args.gpu}))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        os.environ["HIP_VISIBLE_DEVICES"] = args.gpu  # the reference sets CUDA_VISIBLE_DEVICES (:275)
    import torch
    from video_dqn_amd.config import ExperimentConfig
    from video_dqn_amd.trainer import run_train

    if os.environ.get("VDQN_SINGLE_DEVICE") == "1":  # functional test of the N > 1 path on a 1-GPU box (with gloo)
        local_rank = 0
    device = f"cuda:{local_rank}" if world > 1 else "cuda"
    if world > 1:
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VDQN_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" for functional tests on one GPU
        torch.distributed.init_process_group(backend, rank=rank, world_size=world,
                                             **({"device_id": torch.device(device)} if backend == "nccl" else {}))
    config = ExperimentConfig(args.config, device=device, remove=args.delete and rank == 0, resume=args.resume or rank != 0,
                              tensorboard=(rank == 0))
    if rank == 0:
        with open(f"{config.folder}/log", "w") as text_file:  # :283-284
            text_file.write(f"Running with config ({str(config.cfg)})")
    resume_from = -1
    if args.resume:  # :286-294
        mdir = f"{config.folder}/models"
        models = [n for n in (os.listdir(mdir) if os.path.isdir(mdir) else []) if n.startswith("sample") and n.endswith(".torch")]
        if models:
            resume_from = max(int(n[6:-6]) for n in models)
            print(f"Resuming from: {resume_from}")
    run_train(config, resume_from, max_steps=args.max_steps, rank=rank, world_size=world)
    if world > 1:
        torch.distributed.destroy_process_group()
