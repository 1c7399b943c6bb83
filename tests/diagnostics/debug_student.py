"""Diagnostic (not part of the product): per-parameter gradient error of the student step."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sfod = importlib.import_module("simple-sfod_amd")
import test_gpu_model as T

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

model, sd, losses, losses_ref = T._student_vs_oracle(sfod, 2, 160, 224, [3, 5], "fp32", 3)
for k in losses_ref: print(k, losses[k].item(), losses_ref[k].item())
for name, p in model.named_parameters():
    if name.startswith("DC_"): continue
    print(f"{name:60s} {rel(p.grad, sd[name].grad):.3e}  |ref|={sd[name].grad.norm().item():.3e}")
