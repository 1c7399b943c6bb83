"""One rank of tests/test_gpu_two_rank.py: BASELINE config #4's code path (image-parallel teacher-student step,
world_size 2) on the HIP kernels, both ranks on ``cuda:0``.

RCCL refuses two ranks on one device; gloo does not (it stages device tensors through the host), and the trainer only
talks to ``torch.distributed`` -- so everything a real 2-GPU run executes runs here: the constructor's broadcast of
rank 0's state, the rank-strided loader shard with per-rank batch ``IMS_PER_BATCH_TARGET // 2``, the three phases of
``GradientReducer`` launched from inside the backbone's backward, the 1/world factor in the fused SGD kernel, per-rank
EMA / BatchNorm statistics, the rank-mean metrics flush.  Started as a fresh interpreter by the test (never forked,
never re-exec'd); writes what it saw to ``<out>/rank<r>.pt`` for the parent, which runs the oracle.
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vgg")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--out", required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=384)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{}".format(args.port), rank=rank, world_size=world)
    info = {"rank": rank, "world": world, "gloo_device_tensors": True}
    try:
        probe = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(probe)
        assert probe[0].item() == sum(range(1, world + 1))
    except (RuntimeError, NotImplementedError) as e:      # a gloo build without device support: stage through the host here
        info["gloo_device_tensors"] = False
        info["gloo_error"] = repr(e)
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
    resnet = args.model == "r101"
    H, W = args.height, args.width
    yaml = os.path.join(ROOT, "configs", "r101_c4_cs_foggy_adaptive_teacher_source_free.yaml" if resnet
                        else "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")
    cfg = sfod.config.setup_cfg(yaml, [
        "OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", args.dtype, "SOLVER.IMS_PER_BATCH_TARGET", str(world),      # B = 1 per rank
        "SFOD.SYNTHETIC.HEIGHT", str(H), "SFOD.SYNTHETIC.WIDTH", str(W), "SFOD.SYNTHETIC.NUM_IMAGES", "8",
        "INPUT.MIN_SIZE_TRAIN", "({},)".format(H), "INPUT.RANDOM_FLIP", "none", "SOLVER.WARMUP_ITERS", "0",
        "SOLVER.BASE_LR", "2.5e-5", "SFOD.EMA.KEEP_RATE", "0.9", "SOLVER.CHECKPOINT_PERIOD", "0", "SEED", "3",
        "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False", "SFOD.EVAL_HOOK", "False", "SFOD.ELIDE_DEAD_BRANCHES", "True",
        "SFOD.OVERLAP_TEACHER", "True", "WEAK_STRONG_AUGMENT", "False", "SFOD.DETERMINISTIC", "True"])
    # per-rank RNG state at construction (the reference seeds SEED + rank): the draws DIFFER, the constructor's broadcast
    # is what makes the ranks start from rank 0's weights
    torch.manual_seed(100 + rank)
    info["rng_probe"] = torch.randn(4)
    torch.manual_seed(100 + rank)
    tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
    info["batch_per_rank"] = tr.data_loader.batch
    red = tr._reducer
    info["reducer"] = None if red is None else {"early": (red.lo, red.hi), "mid": (red.mlo, red.mhi), "skip": list(red.skip),
                                                "final_elements": red.final_elements()}
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    info["init_student"] = cpu(tr.model.state_dict())
    with torch.no_grad():       # planted labels + a student 2 % off the teacher (tests/test_gpu_trajectory.py), same on both ranks
        tr.model.roi_heads.box_predictor.cls_score.weight.mul_(4.0 if resnet else 30.0)
        tr._copy_main_model()
        gen = torch.Generator(device="cuda").manual_seed(11)
        for n, p in tr.model.named_parameters():
            if not n.startswith("DC_"):
                p.mul_(1.0 + 0.02 * torch.randn(p.shape, device="cuda", generator=gen))
    A = 12 if resnet else 15
    Hf, Wf = (-(-H // 16), -(-W // 16)) if resnet else (H // 32, W // 32)
    g = torch.Generator().manual_seed(1 + rank)       # every rank samples with its own RNG
    rpn_keys = torch.randint(0, 2 ** 31 - 1, (1, Hf * Wf * A), generator=g, dtype=torch.int64)
    roi_keys = torch.randint(0, 2 ** 31 - 1, (1, 2100), generator=g, dtype=torch.int64)
    tr.model.proposal_generator._forced_keys = rpn_keys.to(torch.int32).cuda()
    tr.model.roi_heads._forced_keys = roi_keys.to(torch.int32).cuda()
    cap = {}
    s_rpn = tr.model.proposal_generator
    orig_props, orig_teacher = s_rpn._proposals, tr._teacher_pass

    def cap_props(*a, **k):
        p = orig_props(*a, **k)
        cap.setdefault("props", []).append(p)
        return p

    def cap_teacher(data_k):
        cap["images"] = [d["image"].cpu().clone() for d in data_k]
        cap["image_ids"] = [int(d["image_id"]) for d in data_k]
        cap["pseudo"] = orig_teacher(data_k)
        return cap["pseudo"]
    s_rpn._proposals, tr._teacher_pass = cap_props, cap_teacher
    # the reducer's phases as they are launched from inside the backward
    launched = []
    if red is not None:
        for name in ("launch_early", "launch_mid", "finish"):
            orig = getattr(red, name)

            def wrapped(_orig=orig, _name=name):
                pending = (red.work is not None, red.work_mid is not None)
                _orig()
                launched.append((_name, pending, (red.work is not None, red.work_mid is not None)))
            setattr(red, name, wrapped)
        bb = tr.model.backbone
        bb._pre_backward = red.launch_early
        if bb._mid_backward is not None:
            bb._mid_backward = red.launch_mid
    info["before_student"], info["before_teacher"] = cpu(tr.model.state_dict()), cpu(tr.model_teacher.state_dict())
    tr.iter = 0
    tr.run_step()
    local_metrics = {k: (float(v) if not isinstance(v, torch.Tensor) else float(v.item())) for k, v in tr.storage._pending.items()}
    tr.after_step()
    rec = tr._flush_metrics() if tr.storage._pending else tr.storage.history[-1]
    torch.cuda.synchronize()
    f = tr.optimizer.flat
    info.update({
        "after_student": cpu(tr.model.state_dict()), "after_teacher": cpu(tr.model_teacher.state_dict()),
        "momentum": {n: tr.optimizer.mom[o:o + k].view(shp).cpu().clone() for n, (o, k, shp) in f.offsets.items()},
        "grad_sum": f.grad[: f.n_norm_end].double().abs().sum().item(), "grad_scale": tr.optimizer.grad_scale,
        "dc_grad_abs": sum(f.grad[o:o + k].abs().sum().item() for n, (o, k, _) in f.offsets.items() if n.startswith("DC_")),
        "record": rec, "local_metrics": local_metrics, "launched": launched,
        "images": cap["images"], "image_ids": cap["image_ids"],
        "pseudo_boxes": [cap["pseudo"].boxes[0, : cap["pseudo"].count[0].item()].cpu()],
        "pseudo_classes": [cap["pseudo"].classes[0, : cap["pseudo"].count[0].item()].cpu().long()],
        "props": [(cap["props"][0].boxes[0, : cap["props"][0].count[0].item()].cpu(),
                   cap["props"][0].logits[0, : cap["props"][0].count[0].item()].cpu())],
        "rpn_keys": rpn_keys, "roi_keys": roi_keys,
        "names": [n for n, p_ in tr.model.named_parameters() if p_.requires_grad],
        "frozen": [n for n, p_ in tr.model.named_parameters() if not p_.requires_grad],
        "dc_on": bool(cfg.DOMAIN_CLASSIFIER.ENABLED), "elided_bn_updates": tr._elided_bn_updates,
    })
    # next shard element, to show the stride of the sampler beyond one step
    nxt = next(tr._data_loader_iter)
    info["next_image_ids"] = [int(d["image_id"]) for d in nxt[1]]
    torch.save(info, os.path.join(args.out, "rank{}.pt".format(rank)))
    dist.barrier()
    dist.destroy_process_group()
    sfod.native.set_deterministic(False)


if __name__ == "__main__":
    main()
