"""A rank of the launcher test: what bench.py / train_net_mt.py do after the self-launch, on CPU (gloo):
read RANK / WORLD_SIZE / MASTER_* from the environment, form the group, run the step's one collective."""
import argparse
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _launcher():
    spec = importlib.util.spec_from_file_location("sfod_launch", os.path.join(ROOT, "simple-sfod_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--fail-rank", type=int, default=-1)
    args = ap.parse_args()
    lm = _launcher()
    if args.gpus > 1 and not lm.under_launcher():      # the same two lines as bench.py / train_net_mt.py
        assert "torch" not in sys.modules, "the parent must not have imported torch before launching"
        sys.exit(lm.launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    assert world == args.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if rank == args.fail_rank:
        sys.exit(3)
    dist.init_process_group("gloo")
    sys.path.insert(0, ROOT)
    import importlib
    sfod = importlib.import_module("simple-sfod_amd")
    # the trainer constructor's DDP-style broadcast: rank-dependent initial state -> rank 0's everywhere
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    model[1].running_mean.fill_(float(rank))
    model[1].num_batches_tracked.fill_(7 + rank)
    flat = sfod.engine.FlatModelState(model)
    from types import SimpleNamespace
    sfod.engine.trainer.BaseTrainer._broadcast_initial_state(SimpleNamespace(optimizer=SimpleNamespace(flat=flat)))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    chk = flat.param.double().sum().item()
    parts = [None] * world
    dist.all_gather_object(parts, (chk, flat.fbuf.tolist(), flat.ibuf.tolist()))
    if rank == 0:
        with open(args.out, "w") as f:
            json.dump({"world": world, "sum": t.item(), "states_equal": all(p == parts[0] for p in parts),
                       "running_mean0": flat.fbuf[0].item(), "nbt": flat.ibuf.tolist()}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
