"""Where the HOST spends its time enqueueing a step (cProfile over the steps of bench.py's trainer; the GPU runs behind).
   python tools/experiments/host_profile.py [--model r101] [--batch 1] [--steps 40]"""
import argparse, cProfile, importlib, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vgg"); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
sys.argv = ["bench.py"]
bench = importlib.import_module("bench")
sfod = importlib.import_module("simple-sfod_amd"); sfod.native.load()
args = argparse.Namespace(model=a.model, batch=a.batch, trainer="source_free", no_overlap=False, res="r600", opts=[], host_frames=False)
cfg, tr = bench.build_trainer(sfod, args, bench.PARITY_DTYPE[a.model], 1, 0, 0)
sfod.engine.planted.plant_labels(tr, sfod.engine.planted.SCALE[a.model], note=lambda m: None, bias=None)
bench.run_steps(tr, 0, 10); torch.cuda.synchronize()
tr._max_in_flight = 0                      # let the host run ahead: its own pace is what is measured
pr = cProfile.Profile(); pr.enable()
bench.run_steps(tr, 10, a.steps)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats("tottime").print_stats(38)
print(s.getvalue()[:9000])
