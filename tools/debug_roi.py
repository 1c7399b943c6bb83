"""Diagnostic: ROI-head chain stage by stage against an fp64 CPU chain on identical inputs."""
import importlib, os, sys, math
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sfod = importlib.import_module("simple-sfod_amd")
N = sfod.native
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
g = torch.Generator().manual_seed(0)
R, K1, D = 1024, 25088, 1024
x0 = torch.relu(torch.randn(R, K1, generator=g)); x0[525:] = 0
W1 = (torch.rand(D, K1, generator=g) * 2 - 1) * math.sqrt(3 / K1)
W2 = (torch.rand(D, D, generator=g) * 2 - 1) * math.sqrt(3 / D)
Wp = torch.randn(41, D, generator=g) * 0.01
dpred = torch.zeros(R, 48); dpred[:525, :41] = torch.randn(525, 41, generator=g) * 1e-3
# fp64 truth
x0d, W1d, W2d, Wpd, dpd = [t.double() for t in (x0, W1, W2, Wp, dpred)]
h1 = torch.relu(x0d @ W1d.T); h2 = torch.relu(h1 @ W2d.T); pred = h2 @ Wpd.T
dh2 = (dpd[:, :41] @ Wpd) * (h2 > 0); dW2 = dh2.T @ h1; dh1 = (dh2 @ W2d) * (h1 > 0); dW1 = dh1.T @ x0d; dx0 = dh1 @ W1d
# fp32 torch CPU
h1f = torch.relu(x0 @ W1.T); h2f = torch.relu(h1f @ W2.T)
dh2f = (dpred[:, :41] @ Wp) * (h2f > 0); dW2f = dh2f.T @ h1f; dh1f = (dh2f @ W2) * (h1f > 0); dW1f = dh1f.T @ x0
print("cpu-fp32 vs fp64: h1 %.2e h2 %.2e dh2 %.2e dW2 %.2e dh1 %.2e dW1 %.2e" % (rel(h1f, h1), rel(h2f, h2), rel(dh2f, dh2), rel(dW2f, dW2), rel(dh1f, dh1), rel(dW1f, dW1)))
# GPU
dev = "cuda"
xg = x0.to(dev); dt = N.F32
w1p = N.pack_fc_weight(W1.to(dev), dt); h1g = N.conv_fwd(xg, w1p, None, D, 1, act=1)
w2p = N.pack_fc_weight(W2.to(dev), dt); h2g = N.conv_fwd(h1g, w2p, None, D, 1, act=1)
print("gpu h1 %.2e h2 %.2e  mask flips h1 %d h2 %d" % (rel(h1g, h1), rel(h2g, h2), int(((h1g.cpu() > 0) != (h1 > 0)).sum()), int(((h2g.cpu() > 0) != (h2 > 0)).sum())))
dpg = dpred.to(dev)
wpt = N.pack_fc_weight(Wp.to(dev), dt, transpose=True, ld=48)
dh2g = N.conv_fwd(dpg, wpt, None, D, 1); print("gpu dh2 premask %.2e" % rel(dh2g, dpd[:, :41] @ Wpd))
N.act_bwd_(dh2g, h2g, 1); print("gpu dh2 %.2e" % rel(dh2g, dh2))
dW2g = N.conv_wgrad(h1g, dh2g, D, 1).view(D, D); print("gpu dW2 %.2e" % rel(dW2g, dW2))
dW2x = N.conv_wgrad(h1.float().to(dev), dh2.float().to(dev), D, 1).view(D, D); print("gpu dW2 (exact inputs) %.2e" % rel(dW2x, dW2))
w2t = N.pack_fc_weight(W2.to(dev), dt, transpose=True)
dh1g = N.conv_fwd(dh2g, w2t, None, D, 1); print("gpu dh1 premask %.2e" % rel(dh1g, dh2 @ W2d))
dh1x = N.conv_fwd(dh2.float().to(dev), w2t, None, D, 1); print("gpu dh1 premask (exact inputs) %.2e" % rel(dh1x, dh2 @ W2d))
N.act_bwd_(dh1g, h1g, 1); print("gpu dh1 %.2e" % rel(dh1g, dh1))
dW1g = N.conv_wgrad(xg, dh1g, D, 1).view(D, K1); print("gpu dW1 %.2e" % rel(dW1g, dW1))
w1t = N.pack_fc_weight(W1.to(dev), dt, transpose=True)
dx0g = N.conv_fwd(dh1g, w1t, None, K1, 1); print("gpu dx0 %.2e" % rel(dx0g, dx0))
