"""Host-side (Python) profile of trainer steps: where the launch-bound configs spend their time.
  python tools/profile_host.py [--model r101] [--batch 2] [--steps 5]"""
import argparse, cProfile, importlib, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="r101")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dtype", default="bf16x3")
args = ap.parse_args()
sfod = importlib.import_module("simple-sfod_amd")
yaml = {"vgg": "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml",
        "r101": "r101_c4_cs_foggy_adaptive_teacher_source_free.yaml"}[args.model]
cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs", yaml),
                            ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", args.dtype, "SOLVER.IMS_PER_BATCH_TARGET", str(args.batch),
                             "SOLVER.CHECKPOINT_PERIOD", "0", "SFOD.SYNTHETIC.NUM_IMAGES", "16", "MODEL.DEVICE", "cuda:0"])
tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
for i in range(3):
    tr.iter = i; tr.run_step(); tr.scheduler.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(args.steps):
    tr.iter = 3 + i; tr.run_step(); tr.scheduler.step()
pr.disable()
import time
t_host = time.perf_counter()
torch.cuda.synchronize()
print("host finished %.1f ms before the GPU (0 = host-bound)" % ((time.perf_counter() - t_host) * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
st.print_callers("method 'to'")
st.print_callers("method 'item'|method 'tolist'|method 'cpu'")
