"""N>1 path on CPU: two gloo ranks exercise the sharding rule (TrainingSampler striding, per-rank
batch = IMS_PER_BATCH_TARGET // world) and the one collective of the step (sum all-reduce of the
flat gradient buffer, 1/world folded into the optimiser) -- build.py:337-343, SURVEY.md section 8e."""
import importlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sfod = importlib.import_module("simple-sfod_amd")
    from types import SimpleNamespace
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(5, 3))
    flat = sfod.engine.FlatModelState(model)
    # parameters are views of the flat buffer, norm params grouped after the decayed ones
    names = [n for n, _ in flat.order]
    assert names == ["0.weight", "0.bias", "2.weight", "2.bias", "1.weight", "1.bias"]
    for n, p in flat.order:
        o, k, shp = flat.offsets[n]
        assert p.data_ptr() == flat.param.data_ptr() + 4 * o
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
    # rank-dependent gradients through autograd land in the flat buffer
    x = torch.full((2, 3, 7, 7), float(rank + 1))
    y = model[1](model[0](x)).sum() + model[2](torch.ones(1, 5) * (rank + 1)).sum()
    y.backward()
    local = flat.grad.clone()
    opt = SimpleNamespace(flat=flat, grad_scale=1.0)
    tr = SimpleNamespace(optimizer=opt)
    sfod.engine.trainer.BaseTrainer._reduce_gradients(tr)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok_sum = torch.allclose(flat.grad, sum(gathered))
    # two-phase reducer: the "heads" slice (here module 2) goes first and asynchronously, the rest afterwards
    flat.grad.copy_(local)
    red = sfod.engine.trainer.GradientReducer(flat, prefixes=("2.",))
    o2, k2, _ = flat.offsets["2.weight"]
    assert red.lo == o2 and red.hi >= flat.offsets["2.bias"][0] + flat.offsets["2.bias"][1]
    red.launch_early()
    tr2 = SimpleNamespace(optimizer=opt, _reducer=red)
    sfod.engine.trainer.BaseTrainer._reduce_gradients(tr2)
    ok_sum = ok_sum and torch.allclose(flat.grad, sum(gathered)) and red.work is None
    # three phases: module 2 early, module 0 (the "deep trunk" slice) mid-backward, the norm parameters at the end
    flat.grad.copy_(local)
    red3 = sfod.engine.trainer.GradientReducer(flat, prefixes=("2.",), mid_prefixes=("0.",))
    assert red3.mlo == flat.offsets["0.weight"][0] and red3.mhi == flat.offsets["2.weight"][0]
    red3.launch_early()
    red3.launch_mid()
    tr3 = SimpleNamespace(optimizer=opt, _reducer=red3)
    sfod.engine.trainer.BaseTrainer._reduce_gradients(tr3)
    ok_sum = ok_sum and torch.allclose(flat.grad, sum(gathered)) and red3.work is None and red3.work_mid is None
    # only the mid phase was launched (e.g. no heads slice): the rest is reduced in finish()
    flat.grad.copy_(local)
    red4 = sfod.engine.trainer.GradientReducer(flat, prefixes=("nothing.",), mid_prefixes=("0.",))
    red4.launch_early()
    red4.launch_mid()
    sfod.engine.trainer.BaseTrainer._reduce_gradients(SimpleNamespace(optimizer=opt, _reducer=red4))
    ok_sum = ok_sum and torch.allclose(flat.grad, sum(gathered))
    # prefixes that match scattered parameters: the longest pure run is used (0.weight, 108 elements, vs 2.*, 18)
    red_run = sfod.engine.trainer.GradientReducer(flat, prefixes=("0.weight", "2."), mid_prefixes=())
    ok_sum = ok_sum and red_run.lo == flat.offsets["0.weight"][0] and red_run.hi == flat.offsets["0.bias"][0]
    # the real model's parameter layout (modules constructed on the CPU: no kernel runs at construction): the three
    # phases cover disjoint slices, every element is reduced exactly once, result == plain all-reduce
    cfg_m = sfod.config.setup_cfg(os.path.join(ROOT, "configs",
                                               "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"), [])
    torch.manual_seed(1)
    big = sfod.registry.META_ARCH_REGISTRY.get(cfg_m.MODEL.META_ARCHITECTURE)(cfg_m)
    bflat = sfod.engine.FlatModelState(big, frozen_prefixes=("DC_img.", "DC_ins."))
    g = torch.Generator().manual_seed(100 + rank)
    bflat.grad.copy_(torch.randn(bflat.grad.numel(), generator=g))
    bflat.grad[bflat.n_norm_end:].zero_()     # not optimised (here: the domain classifier): no gradient is ever written, none exchanged
    mine_g = bflat.grad.clone()
    parts = [torch.zeros_like(mine_g) for _ in range(world)]
    dist.all_gather(parts, mine_g)
    redm = sfod.engine.trainer.GradientReducer(bflat)
    n_heads, n_mid = redm.hi - redm.lo, redm.mhi - redm.mlo
    ok_model = n_heads > 25_000_000 and n_mid > 14_000_000 and redm.mhi <= redm.lo
    redm.launch_early()
    redm.launch_mid()
    sfod.engine.trainer.BaseTrainer._reduce_gradients(SimpleNamespace(optimizer=SimpleNamespace(flat=bflat, grad_scale=1.0),
                                                                      _reducer=redm))
    ok_sum = ok_sum and ok_model and torch.equal(bflat.grad, sum(parts))
    del big, bflat, parts
    # BASELINE config #5 (an 8-GPU config): the ResNet-101-C4 layout.  The backbone names res4 for the mid phase; heads,
    # res4 and the rest are disjoint, every element is reduced exactly once, and the blocking rest is res3 (< 10 MB)
    cfg_r = sfod.config.setup_cfg(os.path.join(ROOT, "configs", "r101_c4_cs_foggy_adaptive_teacher_source_free.yaml"), [])
    torch.manual_seed(1)
    r101 = sfod.registry.META_ARCH_REGISTRY.get(cfg_r.MODEL.META_ARCHITECTURE)(cfg_r)
    rflat = sfod.engine.FlatModelState(r101, frozen_prefixes=("DC_img.", "DC_ins."))
    g = torch.Generator().manual_seed(200 + rank)
    rflat.grad.copy_(torch.randn(rflat.grad.numel(), generator=g))
    rparts = [torch.zeros_like(rflat.grad) for _ in range(world)]
    dist.all_gather(rparts, rflat.grad.clone())
    mids = r101.backbone.reduce_schedule(2 * 600 * 1200)
    redr = sfod.engine.trainer.GradientReducer(rflat, mid_prefixes=mids)
    live = sum(p_.numel() for p_ in r101.backbone.parameters() if p_.requires_grad)
    # (buffer order: weights, norm parameters, then everything that is not optimised -- the disabled domain classifier here)
    rest = redr.final_elements()
    ok_r101 = (tuple(mids) == ("backbone.res4.",) and (redr.mhi - redr.mlo) > 0.94 * live and redr.mhi <= redr.lo
               and (redr.hi - redr.lo) > 100_000_000 and 0 < 4 * rest < 10_000_000 and rflat.n_norm_end < rflat.grad.numel())
    rflat.grad[rflat.n_norm_end:].zero_()            # no gradient is ever written there
    for t in rparts:
        t[rflat.n_norm_end:].zero_()
    redr.launch_early()
    redr.launch_mid()
    sfod.engine.trainer.BaseTrainer._reduce_gradients(SimpleNamespace(optimizer=SimpleNamespace(flat=rflat, grad_scale=1.0),
                                                                      _reducer=redr))
    ok_sum = ok_sum and ok_r101 and torch.equal(rflat.grad, sum(rparts))
    del r101, rflat, rparts
    # logged metrics: mean over ranks of the period's device scalars with one small all-reduce, data_time = max over
    # ranks (reference base.py:198-209: comm.gather + np.mean / np.max)
    st = sfod.engine.trainer.EventStorage(0)
    st.put_scalars(True, loss_cls=torch.tensor(float(rank + 1)), total_loss=torch.tensor(10.0 * (rank + 1)),
                   data_time=0.25 * (rank + 1), **{"lr_note": 3.0 * (rank + 1), "rank_mean": torch.tensor(2.0 * (rank + 1))})
    # what a hook or the teacher pass puts beside the step's metrics_dict stays rank-local (the reference logs the main
    # process's storage: val_loss.py:64-66, source_free_adaptive_teacher.py:411-423)
    st.put_scalar("roi_head/mean_confidence", torch.tensor(0.5 + rank))
    st.put_scalar("total_loss_val", torch.tensor(7.0 * (rank + 1)))
    rec = st.flush(reduce_over_ranks=True)
    ok_sum = ok_sum and rec["loss_cls"] == 1.5 and rec["total_loss"] == 15.0 and rec["data_time"] == 0.5 \
        and rec["lr_note"] == 4.5 and rec["rank_mean"] == 3.0 and rec["roi_head/mean_confidence"] == 0.5 + rank \
        and rec["total_loss_val"] == 7.0 * (rank + 1) and st.flush(reduce_over_ranks=True) == {}
    # rank-local keys may differ in number between ranks (an empty share of a tiny evaluation set): they are never part
    # of a collective, the rank-mean keys are still averaged
    st.put_scalars(True, a=torch.tensor(float(rank)))
    st.put_scalars(**({"b": torch.tensor(5.0)} if rank == 0 else {}))
    rec = st.flush(reduce_over_ranks=True)
    ok_sum = ok_sum and rec["a"] == 0.5 and (("b" in rec) == (rank == 0))
    # ranks that disagree on the number of rank-mean keys: no mismatched collective, every rank keeps its own values
    st.put_scalars(True, a=torch.tensor(float(rank)), **({"c": torch.tensor(5.0)} if rank == 0 else {}))
    rec = st.flush(reduce_over_ranks=True)
    ok_sum = ok_sum and rec["a"] == float(rank) and (("c" in rec) == (rank == 0))
    # the numbers the REFERENCE's _write_metrics logged for two ranks (tests/golden/glue_ref.npz, bt_*: base.py:186-220 run with
    # comm.gather returning both ranks' dicts): the product's storage reproduces them rank by rank
    import numpy as np
    fxg = np.load(os.path.join(ROOT, "tests", "golden", "glue_ref.npz"), allow_pickle=False)
    mk = [str(k) for k in fxg["bt_metric_keys"]]
    mine_vals = dict(zip(mk, (fxg["bt_rank0"] if rank == 0 else fxg["bt_rank1"]).tolist()))
    holder = SimpleNamespace(storage=sfod.engine.trainer.EventStorage(0))
    sfod.engine.trainer.BaseTrainer._write_metrics(holder, {k: (torch.tensor(v) if k != "data_time" else v) for k, v in mine_vals.items()})
    rec = holder.storage.flush(reduce_over_ranks=True)
    logged = dict(zip([str(k) for k in fxg["bt_logged_keys"]], fxg["bt_logged_vals"].tolist()))
    ok_sum = ok_sum and sorted(k for k in rec if k != "iteration") == sorted(logged) and all(abs(rec[k] - v) < 1e-6 for k, v in logged.items())
    # sampler: rank r takes elements r, r+W, ... of ONE shared-seed stream
    s = iter(sfod.data.TrainingSampler(10, seed=7, rank=rank, world=world))
    mine = [next(s) for _ in range(10)]
    full = iter(sfod.data.TrainingSampler(10, seed=7, rank=0, world=1))
    stream = [next(full) for _ in range(20)]
    ok_sampler = mine == stream[rank::world]
    # loader: per-rank batch and divisibility assert
    cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs",
                                             "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"),
                                ["SOLVER.IMS_PER_BATCH_TARGET", "4", "SFOD.SYNTHETIC.HEIGHT", "64",
                                 "SFOD.SYNTHETIC.WIDTH", "96", "SFOD.SYNTHETIC.NUM_IMAGES", "6",
                                 "INPUT.MIN_SIZE_TRAIN", "(32,)", "SFOD.SYNTHETIC.BOXES_PER_IMAGE", "2"])
    loader = sfod.data.TwoCropLoader(cfg, torch.device("cpu"), rank, world)
    strong, weak = next(loader)
    ids = [d["image_id"] for d in weak]
    all_ids = [None] * world
    dist.all_gather_object(all_ids, ids)
    ok_loader = len(weak) == 2 and weak[0]["image"].dtype == torch.uint8 and tuple(weak[0]["image"].shape) == (3, 32, 48)
    bad = cfg.clone()
    bad.defrost()
    bad.SOLVER.IMS_PER_BATCH_TARGET = 3
    try:
        sfod.data.TwoCropLoader(bad, torch.device("cpu"), rank, world)
        ok_assert = False
    except AssertionError:
        ok_assert = True
    q.put((rank, ok_sum, opt.grad_scale, ok_sampler, ok_loader, ok_assert, all_ids))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_gradient_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_sum, gscale, ok_sampler, ok_loader, ok_assert, all_ids in res:
        assert ok_sum and ok_sampler and ok_loader and ok_assert, (rank, ok_sum, ok_sampler, ok_loader, ok_assert)
        assert gscale == 0.5
    # the two ranks drew disjoint strided slices of the same permutation
    ids = res[0][6]
    assert len(set(ids[0]) & set(ids[1])) == 0 or len(set(ids[0] + ids[1])) <= 6


def _eval_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sfod = importlib.import_module("simple-sfod_amd")
    S = sfod.structures
    cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs",
                                             "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"),
                                ["MODEL.DEVICE", "cpu", "SFOD.SYNTHETIC.HEIGHT", "64", "SFOD.SYNTHETIC.WIDTH", "128",
                                 "SFOD.SYNTHETIC.NUM_TEST_IMAGES", "5", "TEST.IMS_PER_BATCH", "2", "INPUT.MIN_SIZE_TEST",
                                 "32", "SFOD.SYNTHETIC.BOXES_PER_IMAGE", "4",
                                 "DATASETS.TEST", "('synthetic_cityscapes_foggy_val',)"])

    class Echo(torch.nn.Module):
        """Returns half of each image's ground truth (even indices) as detections: AP is < 100 and depends on
        every image, so a rank that dropped or duplicated its share would change the table."""

        def forward(self, batched_inputs):
            outs = []
            for d in batched_inputs:
                inst = S.Instances((d["height"], d["width"]))
                sx = d["width"] / d["image"].shape[2]
                keep = torch.arange(0, len(d["instances"]), 2)
                inst.pred_boxes = S.Boxes(d["instances"].gt_boxes.tensor[keep] * sx)
                inst.scores = torch.linspace(0.9, 0.6, len(keep))
                inst.pred_classes = d["instances"].gt_classes[keep]
                outs.append({"instances": inst})
            return outs

    # rank-local share (InferenceSampler) + gather in evaluate(): rank 0 gets the table, the others {}
    res = sfod.engine.BaseTrainer.test(cfg, Echo())
    loader = sfod.data.TestLoader(cfg, torch.device("cpu"), rank, world)
    ids = [d["image_id"] for b in loader for d in b]
    q.put((rank, res, ids))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_evaluation_gathers_predictions_on_rank0(sfod):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        rank, r, ids = q.get(timeout=300)
        res[rank] = (r, ids)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]
    assert res[1][0] == {}                                   # d2: only the main process evaluates
    table = res[0][0]["bbox"]
    # single-process reference over the whole set
    cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs",
                                             "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"),
                                ["MODEL.DEVICE", "cpu", "SFOD.SYNTHETIC.HEIGHT", "64", "SFOD.SYNTHETIC.WIDTH", "128",
                                 "SFOD.SYNTHETIC.NUM_TEST_IMAGES", "5", "TEST.IMS_PER_BATCH", "2", "INPUT.MIN_SIZE_TEST",
                                 "32", "SFOD.SYNTHETIC.BOXES_PER_IMAGE", "4"])
    loader = sfod.data.TestLoader(cfg, torch.device("cpu"))
    ev = sfod.engine.BaseTrainer.build_evaluator(cfg, "x", data_loader=loader)
    S = sfod.structures
    for batch in loader:
        outs = []
        for d in batch:
            inst = S.Instances((d["height"], d["width"]))
            keep = torch.arange(0, len(d["instances"]), 2)
            inst.pred_boxes = S.Boxes(d["instances"].gt_boxes.tensor[keep] * (d["width"] / d["image"].shape[2]))
            inst.scores = torch.linspace(0.9, 0.6, len(keep))
            inst.pred_classes = d["instances"].gt_classes[keep]
            outs.append({"instances": inst})
        ev.process(batch, outs)
    ref = ev.evaluate()["bbox"]
    assert 0 < ref["AP"] < 100
    for k, v in ref.items():
        assert (v != v and table[k] != table[k]) or abs(table[k] - v) < 1e-9, k


def test_self_launching_entry_point_starts_two_gloo_ranks(tmp_path):
    """``python <script> --gpus 2`` without a launcher: the script starts its ranks itself through
    simple-sfod_amd/launch.py (what bench.py --gpus N and train_net_mt.py --num-gpus N do, reference
    train_net_mt.py:90-101) -- the parent never imports torch; the ranks find RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1,
    form the group, and the trainer's constructor broadcast makes rank 0's initial state everybody's."""
    import json
    import subprocess
    out = tmp_path / "r.json"
    worker = os.path.join(ROOT, "tests", "helpers", "launch_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    rc = subprocess.call([sys.executable, worker, "--gpus", "2", "--out", str(out)], env=env, timeout=300)
    assert rc == 0
    r = json.loads(out.read_text())
    assert r["world"] == 2 and r["sum"] == 3.0
    assert r["states_equal"] and r["running_mean0"] == 0.0 and r["nbt"][0] == 7      # rank 0's buffers everywhere
    # a failing rank fails the launch (exit code propagates to the caller)
    rc = subprocess.call([sys.executable, worker, "--gpus", "2", "--out", str(out), "--fail-rank", "1"], env=env, timeout=300)
    assert rc != 0


def test_launch_command_line():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sfod_launch", os.path.join(ROOT, "simple-sfod_amd", "launch.py"))
    lm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lm)
    cmd = lm.launch_command("bench.py", ["--gpus", "8", "--steps", "5"], 8, port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                       "--master-port", "29511", "bench.py", "--gpus", "8", "--steps", "5"]
    assert not lm.under_launcher() or "WORLD_SIZE" in os.environ
