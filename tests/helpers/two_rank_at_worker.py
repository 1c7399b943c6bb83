"""One rank of tests/test_gpu_two_rank.py::test_with_source_trainer_two_ranks: ``TRAINER: "adaptive_teacher"`` with world_size 2, both
ranks on ``cuda:0`` over gloo (see two_rank_worker.py for why that exercises the code an N-GPU job runs).  Three steps around
BURN_UP_STEP = 1: burn-in, the hand-over, an EMA step.  Writes what it saw to ``<out>/rank<r>.pt``."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--port", type=int, required=True)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{}".format(args.port), rank=rank, world_size=world)
    try:
        probe = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(probe)
        assert probe[0].item() == sum(range(1, world + 1))
    except (RuntimeError, NotImplementedError):          # a gloo build without device support: stage through the host
        orig_ar, orig_bc = dist.all_reduce, dist.broadcast

        class _Done:
            def wait(self):
                return True

        def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
            h = t.detach().cpu()
            orig_ar(h, op=op)
            t.copy_(h)
            return _Done() if async_op else None

        def broadcast(t, src, group=None, async_op=False):
            h = t.detach().cpu()
            orig_bc(h, src)
            t.copy_(h)
            return _Done() if async_op else None
        dist.all_reduce, dist.broadcast = all_reduce, broadcast
    sfod = importlib.import_module("simple-sfod_amd")
    sfod.native.load()
    cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher.yaml"), [
        "OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", "fp32", "SOLVER.IMS_PER_BATCH", str(world), "SOLVER.IMS_PER_BATCH_TARGET", str(world),
        "SFOD.SYNTHETIC.HEIGHT", "256", "SFOD.SYNTHETIC.WIDTH", "384", "SFOD.SYNTHETIC.NUM_IMAGES", "8", "INPUT.MIN_SIZE_TRAIN", "(192,)",
        "SOLVER.CHECKPOINT_PERIOD", "0", "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False", "SFOD.EVAL_HOOK", "False", "SEED", "3",
        "SEMISUPNET.BURN_UP_STEP", "1", "SEMISUPNET.TEACHER_UPDATE_ITER", "1", "SEMISUPNET.EMA_KEEP_RATE", "0.9",
        "SOLVER.BASE_LR", "0.01", "SFOD.DETERMINISTIC", "True"])
    torch.manual_seed(100 + rank)
    tr = sfod.engine.AdaptiveTeacherTrainer(cfg)
    with torch.no_grad():
        tr.model.roi_heads.box_predictor.cls_score.weight.mul_(60.0)      # (identical on both ranks: after the constructor's broadcast)
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    info = {"rank": rank, "reducer": tr._reducer, "label_batch": tr.data_loader.batch_size_label,
            "unlabel_batch": tr.data_loader.batch_size_unlabel, "ids": [], "students": [], "teachers": [], "recs": []}
    orig_next = tr.data_loader.__next__
    it_ = iter(tr.data_loader)

    class Tap:
        def __iter__(self):
            return self

        def __next__(self):
            b = next(it_)
            info["ids"].append(([int(d["image_id"]) for d in b[1]], [int(d["image_id"]) for d in b[3]]))
            return b
    tr._data_loader_iter = Tap()
    for it in range(3):
        tr.iter = it
        tr.run_step()
        tr.scheduler.step()
        rec = tr._flush_metrics()
        torch.cuda.synchronize()
        info["students"].append(cpu(tr.model.state_dict()))
        info["teachers"].append(cpu(tr.model_teacher.state_dict()))
        info["recs"].append({k: float(v) for k, v in rec.items()})
    torch.save(info, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
