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
