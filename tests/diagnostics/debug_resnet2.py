import importlib, sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet as ore
sfod = importlib.import_module("simple-sfod_amd")
native = sfod.native
rn = importlib.import_module("simple-sfod_amd.modeling.backbone_resnet")
cfg = sfod.config.get_cfg(); sfod.config.add_config(cfg)
cfg.MODEL.RESNETS.DEPTH = 50; cfg.MODEL.RESNETS.NORM = "BN"; cfg.SFOD.COMPUTE_DTYPE = sys.argv[1]
torch.manual_seed(50)
net = rn.ResNet(cfg)
sd = {"backbone." + k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.cuda().train()
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 3, 96, 160, generator=g)
def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
dt = native.F32 if sys.argv[1] == "fp32" else native.BF16
cd = net.compute_dtype
xn = torch.zeros(2, 96, 160, native.chunk_elems(dt), dtype=cd, device="cuda")
xn[..., :3] = x.cuda().permute(0, 2, 3, 1)
with torch.no_grad():
    s = net._stem_forward(xn, dt)
    r = F.conv2d(x, sd["backbone.stem.conv1.weight"], None, stride=2, padding=3)
    r = F.relu(ore.frozen_bn(r, sd, "backbone.stem.conv1.norm"))
    r = F.max_pool2d(r, 3, 2, 1)
    print("stem", rel(s.float().cpu().permute(0, 3, 1, 2), r))
    # res2 block 0 pieces
    blk = net.res2[0]
    b = "backbone.res2.0."
    sc = net._frozen_conv(s, blk.shortcut, 0, dt)
    rsc = ore.frozen_bn(F.conv2d(r, sd[b + "shortcut.weight"]), sd, b + "shortcut.norm")
    print("res2.0.shortcut", rel(sc.float().cpu().permute(0, 3, 1, 2), rsc))
    o1 = net._frozen_conv(s, blk.conv1, 1, dt)
    r1 = F.relu(ore.frozen_bn(F.conv2d(r, sd[b + "conv1.weight"]), sd, b + "conv1.norm"))
    print("res2.0.conv1", rel(o1.float().cpu().permute(0, 3, 1, 2), r1))
    o2 = net._frozen_conv(o1, blk.conv2, 1, dt)
    r2 = F.relu(ore.frozen_bn(F.conv2d(r1, sd[b + "conv2.weight"], padding=1), sd, b + "conv2.norm"))
    print("res2.0.conv2", rel(o2.float().cpu().permute(0, 3, 1, 2), r2))
    o3 = net._frozen_conv(o2, blk.conv3, 0, dt)
    r3 = ore.frozen_bn(F.conv2d(r2, sd[b + "conv3.weight"]), sd, b + "conv3.norm")
    print("res2.0.conv3", rel(o3.float().cpu().permute(0, 3, 1, 2), r3))
    out = native.add_act(o3, sc, 1)
    print("res2.0.out", rel(out.float().cpu().permute(0, 3, 1, 2), F.relu(r3 + rsc)))
