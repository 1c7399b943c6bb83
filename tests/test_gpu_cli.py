"""``train_net_mt.py`` as the user starts it (reference ``train_net_mt.py:34-101``): the script in a child process with a yaml and
``KEY VALUE`` overrides -- trainer dispatch on ``cfg.TRAINER``, the training loop with its periodic checkpoint and ``metrics.json``,
``--resume`` from ``last_checkpoint``, ``--eval-only`` (AdaBN refinement + evaluation, ``:73-82``) -- on small synthetic frames."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["SFOD.SYNTHETIC.HEIGHT", "256", "SFOD.SYNTHETIC.WIDTH", "512", "SFOD.SYNTHETIC.NUM_IMAGES", "6",
         "SFOD.SYNTHETIC.NUM_TEST_IMAGES", "4", "INPUT.MIN_SIZE_TRAIN", "(192,)", "INPUT.MIN_SIZE_TEST", "192",
         "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False", "SFOD.COMPUTE_DTYPE", "bf16x3"]


def _cli(args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_net_mt.py")] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return r.returncode, r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")


def _metrics(out_dir):
    with open(os.path.join(out_dir, "metrics.json")) as fh:
        return [json.loads(line) for line in fh if line.strip()]


@pytest.mark.parametrize("yaml,keys", [
    ("faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml", ("loss_cls_pseudo", "loss_rpn_loc_pseudo")),
    ("faster_rcnn_VGG_cityscapes_source_new.yaml", ("loss_cls", "loss_rpn_loc")),
    ("faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher.yaml", ("loss_cls", "loss_DC_img_s")),
])
def test_cli_trains_checkpoints_and_resumes(tmp_path, yaml, keys):
    out = str(tmp_path / "run")
    base = ["--config-file", os.path.join(ROOT, "configs", yaml)]
    opts = SMALL + ["OUTPUT_DIR", out, "SOLVER.IMS_PER_BATCH", "2", "SOLVER.IMS_PER_BATCH_TARGET", "2", "SOLVER.CHECKPOINT_PERIOD", "2",
                    "SEMISUPNET.BURN_UP_STEP", "1"]
    rc, so, se = _cli(base + opts + ["SOLVER.MAX_ITER", "4"])
    assert rc == 0, se[-3000:]
    assert os.path.exists(os.path.join(out, "model_final.pth")) and os.path.exists(os.path.join(out, "model_0000001.pth"))
    with open(os.path.join(out, "last_checkpoint")) as fh:
        assert fh.read().strip() == "model_final.pth"
    recs = _metrics(out)
    assert recs and all(k in recs[-1] for k in keys) and recs[-1]["iteration"] == 3
    assert all(v == v and abs(v) < 1e6 for r in recs for k, v in r.items() if k.startswith("loss"))
    ck = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert ck["iteration"] == 3
    # --resume: continues at iteration 4 from the files above, up to the new MAX_ITER
    rc, so, se = _cli(base + ["--resume"] + opts + ["SOLVER.MAX_ITER", "6"])
    assert rc == 0, se[-3000:]
    recs2 = _metrics(out)
    assert recs2[-1]["iteration"] == 5 and len(recs2) > len(recs)
    assert torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)["iteration"] == 5


def test_cli_eval_only_runs_the_adabn_refinement_and_prints_the_ap_table(tmp_path):
    out = str(tmp_path / "eval")
    yaml = os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")
    rc, so, se = _cli(["--config-file", yaml, "--eval-only", "--adabn-iters", "3"] + SMALL + ["OUTPUT_DIR", out, "MODEL.WEIGHTS", ""])
    assert rc == 0, se[-3000:]
    assert "bbox" in so and "AP50" in so, so[-2000:]
    assert any(f.startswith("adabn") for f in os.listdir(out)), os.listdir(out)       # base.py:336: the refined model is saved


def _write_cityscapes_like(root, sub, json_name, n, h, w, seed):
    """n PNG frames + a COCO json with the reference's 8 Cityscapes categories (cityscapes-to-coco-conversion/main.py:139-148)"""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    base = os.path.join(root, sub)
    os.makedirs(os.path.join(base, "annotations"), exist_ok=True)
    names = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]
    cats = [{"id": i + 1, "name": nm + "_disk"} for i, nm in enumerate(names)]      # (suffix: the synthetic stand-in has the plain names)
    images, anns, aid = [], [], 1
    for i in range(n):
        hh, ww = (h, w) if i % 3 else (w // 2, h // 2 * 3 // 2)            # mostly landscape, every third frame another shape
        Image.fromarray(rng.integers(0, 256, (hh, ww, 3), dtype=np.uint8), "RGB").save(os.path.join(base, f"f{i}.png"))
        images.append({"id": i + 1, "file_name": f"f{i}.png", "height": hh, "width": ww})
        for _ in range(int(rng.integers(1, 5))):
            bw, bh = float(rng.uniform(16, ww / 2)), float(rng.uniform(16, hh / 2))
            x, y = float(rng.uniform(0, ww - bw)), float(rng.uniform(0, hh - bh))
            anns.append({"id": aid, "image_id": i + 1, "category_id": int(rng.integers(1, 9)), "bbox": [x, y, bw, bh],
                         "iscrowd": 0, "area": bw * bh})
            aid += 1
    with open(os.path.join(base, "annotations", json_name), "w") as fh:
        json.dump({"images": images, "annotations": anns, "categories": cats}, fh)


def test_cli_on_coco_json_files_under_the_references_dataset_names(tmp_path):
    """The reference's yaml names its data ``cityscapes_instancesonly_foggy_train_foggy_beta_0.02`` / ``..._val_...``
    (daod/data/datasets.py:41-108 resolves them under $DETECTRON2_DATASETS): PNG frames of two shapes + COCO json on disk, the
    source-free trainer for a few iterations with the final evaluation on the val split, everything through the CLI."""
    root = str(tmp_path / "datasets")
    _write_cityscapes_like(root, "cityscapes_foggy", "instancesonly_filtered_gtFine_train_foggy_beta_0.02.json", 8, 128, 256, 1)
    _write_cityscapes_like(root, "cityscapes_foggy", "instancesonly_filtered_gtFine_val_foggy_beta_0.02.json", 4, 128, 256, 2)
    out = str(tmp_path / "run")
    yaml = os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")
    env_backup = os.environ.get("DETECTRON2_DATASETS")
    os.environ["DETECTRON2_DATASETS"] = root
    try:
        rc, so, se = _cli(["--config-file", yaml, "OUTPUT_DIR", out, "SOLVER.MAX_ITER", "3", "SOLVER.IMS_PER_BATCH_TARGET", "2",
                           "DATASETS.TRAIN_TARGET", "('cityscapes_instancesonly_foggy_train_foggy_beta_0.02',)",
                           "DATASETS.TEST", "('cityscapes_instancesonly_foggy_val_foggy_beta_0.02',)",
                           "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "256", "INPUT.MIN_SIZE_TEST", "96",
                           "INPUT.MAX_SIZE_TEST", "256", "TEST.VAL_LOSS", "False", "SOLVER.CHECKPOINT_PERIOD", "0",
                           "SFOD.COMPUTE_DTYPE", "bf16x3"])
    finally:
        if env_backup is None:
            del os.environ["DETECTRON2_DATASETS"]
        else:
            os.environ["DETECTRON2_DATASETS"] = env_backup
    assert rc == 0, se[-3000:]
    recs = _metrics(out)
    last = recs[-1]
    assert last["iteration"] == 2 and "loss_cls_pseudo" in last
    ap_keys = [k for k in last if "AP50" in k or k.endswith("/AP")]
    assert ap_keys, sorted(last)                     # the final evaluation's table went into the record (EvalHook after the last step)
    assert any("person_disk" in k for k in last), sorted(last)    # per-class keys with the JSON's category names: the files were read
