import sys, importlib, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
from test_gpu_resnet import _cfg
rn = importlib.import_module("simple-sfod_amd.modeling.backbone_resnet")
def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
dtype = sys.argv[1]
cin, cout, bott, stride = 512, 512, 128, 1
_, cfg = _cfg(50, dtype)
torch.manual_seed(cin + stride)
net = rn.ResNet(cfg).cuda().train()
blk = rn.BottleneckBlock(cin, cout, bott, stride, "BN").cuda().train()
g = torch.Generator().manual_seed(7)
B, H, W = 2, 17, 23
x = torch.randn(B, cin, H, W, generator=g)
ws = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in blk.named_parameters()}
xr = x.clone().requires_grad_(True)
def cbn(t, name, s=1, pad=0):
    y = F.conv2d(t, ws[name + ".weight"], None, stride=s, padding=pad)
    return F.batch_norm(y, None, None, ws[name + ".norm.weight"], ws[name + ".norm.bias"], True, 0.1, 1e-5)
o1 = F.relu(cbn(xr, "conv1", stride)); o1.retain_grad()
o2 = F.relu(cbn(o1, "conv2", 1, 1)); o2.retain_grad()
ref = F.relu(cbn(o2, "conv3") + xr)
w = torch.randn(ref.shape, generator=g)
(ref * w).sum().backward()
xd = x.permute(0, 2, 3, 1).contiguous().cuda()
out, _, sv = net._block_forward(blk, xd, True, native.dt_of_dtype(net.compute_dtype))
print("out", rel(out.cpu().permute(0, 3, 1, 2), ref.detach()))
(xshape, xs, y1, m1, i1, a1, y2, m2, i2, a2, y3, m3, i3, ys, ms, is_, outs) = sv
f32 = lambda t: native.cast(t, torch.float32) if t.dtype != torch.float32 else t
print("a1", rel(f32(a1).cpu().permute(0, 3, 1, 2), o1.detach()), "a2", rel(f32(a2).cpu().permute(0, 3, 1, 2), o2.detach()))
dout = w.permute(0, 2, 3, 1).contiguous().cuda()
gq = native.act_bwd_(dout, outs, 1)
da2, p3 = net._conv_bwd(gq, a2, y3, m3, i3, blk.conv3, False, True)
print("da2", rel(da2.cpu().permute(0, 3, 1, 2), o2.grad))
da1, p2 = net._conv_bwd(da2, a1, y2, m2, i2, blk.conv2, True, True)
print("da1", rel(da1.cpu().permute(0, 3, 1, 2), o1.grad))
dxs, p1 = net._conv_bwd(da1, xs, y1, m1, i1, blk.conv1, True, True)
dx = native.add_(dxs, gq)
print("dx", rel(dx.cpu().permute(0, 3, 1, 2), xr.grad))
