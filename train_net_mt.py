#!/usr/bin/env python
"""CLI drop-in for the reference's ``train_net_mt.py`` (``:34-101``): same config contract
(``--config-file X.yaml [--num-gpus N] [--eval-only] [KEY VALUE ...]``) and trainer dispatch on
``cfg.TRAINER``.  One process per GPU: ``--num-gpus N`` starts the N ranks itself (``train_net_mt.py:90-101`` calls
``detectron2.engine.launch``; here the script re-runs itself under ``python -m torch.distributed.run`` in a child
process, before anything touches the GPU: ``simple-sfod_amd/launch.py``).  Started under an external launcher
(``RANK`` / ``WORLD_SIZE`` in the environment) it is one rank of that job."""
import argparse
import importlib
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _launcher():
    """simple-sfod_amd/launch.py by file path: standard library only, no torch / package import in the parent"""
    spec = importlib.util.spec_from_file_location("sfod_launch", os.path.join(ROOT, "simple-sfod_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--eval-only", action="store_true", help="AdaBN refinement + evaluation (train_net_mt.py:73-82)")
    ap.add_argument("--adabn-iters", type=int, default=1400, help="statistics passes of the refinement (base.py:300)")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    args = ap.parse_args()
    lm = _launcher()
    if args.num_gpus > 1 and not lm.under_launcher():
        sys.exit(lm.launch(os.path.abspath(__file__), sys.argv[1:], args.num_gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.num_gpus and world > 1:
        raise SystemExit(f"--num-gpus {args.num_gpus} but the launcher started {world} ranks")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    sfod = importlib.import_module("simple-sfod_amd")
    cfg = sfod.config.setup_cfg(args.config_file, ["MODEL.DEVICE", f"cuda:{local_rank}"] + args.opts)
    rank = int(os.environ.get("RANK", "0"))
    torch.manual_seed(max(cfg.SEED, 0) + rank)           # d2 default_setup: seed_all_rng(SEED + rank)
    Trainer = sfod.engine.get_trainer_class(cfg)
    sfod.data.register_all_datasets(cfg)                 # train_net_mt.py:71 (names without files -> synthetic stand-in)
    if args.eval_only:
        model = Trainer.build_model(cfg)
        # DetectionCheckpointer(model).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume) (train_net_mt.py:75-77)
        sfod.checkpoint.load_model_weights(model, cfg.MODEL.WEIGHTS)
        loader = sfod.data.TwoCropLoader(cfg, torch.device(cfg.MODEL.DEVICE), rank, world, labeled=True)
        # base.adabn_refinement(cfg, model): reset BN statistics, <= 1400 forward passes, Trainer.test, save "adabn"
        results = sfod.engine.test_refinement(cfg, model, loader, max_iters=args.adabn_iters, trainer_cls=Trainer)
        if rank == 0:
            print(results)
        return results
    trainer = Trainer(cfg)       # N > 1: the constructor broadcasts rank 0's parameters / buffers (DDP semantics)
    if args.resume:                              # the reference keeps this call commented out (train_net_mt.py:86)
        trainer.resume_or_load(resume=True)
    trainer.train()


if __name__ == "__main__":
    main()
