"""Which Python lines of the step launch the small torch kernels (fill / add / copy / cat ...)?
  python tools/profile_torch_ops.py [--model vgg] [--batch 8] [--steps 2]
Prints, per (aten op, innermost frame inside this repo), the number of calls per step."""
import argparse, collections, importlib, os, sys
import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vgg")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=2)
args = ap.parse_args()
sfod = importlib.import_module("simple-sfod_amd")
yaml = {"vgg": "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml",
        "r101": "r101_c4_cs_foggy_adaptive_teacher_source_free.yaml"}[args.model]
cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs", yaml),
                            ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", "bf16x3", "SOLVER.IMS_PER_BATCH_TARGET", str(args.batch),
                             "SOLVER.CHECKPOINT_PERIOD", "0", "SFOD.SYNTHETIC.NUM_IMAGES", "16", "MODEL.DEVICE", "cuda:0"])
tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
for i in range(3):
    tr.iter = i; tr.run_step(); tr.scheduler.step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

agg = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ("view", "permute", "reshape", "detach", "alias", "select", "slice", "expand",
                                       "unsqueeze", "squeeze", "as_strided", "transpose", "t.default", "_unsafe_view",
                                       "empty", "split", "unbind")):
            frame = "<autograd engine / no python frame>"
            for fs in reversed(traceback.extract_stack()):
                if (ROOT in fs.filename) and "profile_torch_ops" not in fs.filename:
                    frame = f"{fs.filename.replace(ROOT + '/', '')}:{fs.lineno} {fs.name}"
                    break
            agg[(name, frame)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    for i in range(args.steps):
        tr.iter = 3 + i; tr.run_step(); tr.scheduler.step()
    torch.cuda.synchronize()
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{n / args.steps:7.1f}/step  {name:34s} {frame}")
