import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet as ore
sfod = importlib.import_module("simple-sfod_amd")
rn = importlib.import_module("simple-sfod_amd.modeling.backbone_resnet")
cfg = sfod.config.get_cfg(); sfod.config.add_config(cfg)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg.MODEL.RESNETS.DEPTH = depth; cfg.MODEL.RESNETS.NORM = "BN"; cfg.SFOD.COMPUTE_DTYPE = sys.argv[2] if len(sys.argv) > 2 else "fp32"
torch.manual_seed(depth)
net = rn.ResNet(cfg)
sd = {"backbone." + k: v.detach().clone() for k, v in net.state_dict().items()}
for k, v in sd.items():
    if v.is_floating_point() and k.startswith(("backbone.res3", "backbone.res4")) and k.endswith(("weight", "bias")):
        v.requires_grad_(True)
net = net.cuda().train()
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 3, 96, 160, generator=g)
ref = ore.forward(sd, x, depth=depth, training=True)
w = torch.randn(ref.shape, generator=g)
(ref * w).sum().backward()
out = net(x.cuda())["res4"]
def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
print("fwd", rel(out.float().cpu(), ref.detach()))
(out.float() * w.cuda()).sum().backward()
for n, p in net.named_parameters():
    if p.grad is None: continue
    print(f"{n:40s} {rel(p.grad.cpu(), sd['backbone.' + n].grad):.3e}")
