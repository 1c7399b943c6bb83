"""Diagnostic: how much do the ORACLE gradients move when every weight is perturbed by EPS (argv[1], default 1e-6;
relative)?  Backbone gradients move by ~1% because a 1e-5 forward difference flips a handful of
ReLU masks / max-pool arg-maxes; this sets the tolerance of the end-to-end gradient parity test.
Measured: feat 2.7e-5, backbone conv grads 0.6-1.5e-2, fc1/fc2 2e-4, RPN head 8e-6."""
import sys, os, torch
EPS = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import model as om
from util_weights import reference_vgg_state
torch.manual_seed(3)
cfg = om.Cfg()
sd0 = om.init_state(cfg, seed=1)
B, H, W = 2, 160, 224
g = torch.Generator().manual_seed(3)
imgs = [torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8) for _ in range(B)]
gtb = [torch.tensor([[10., 20., 120., 100.], [60., 30., 200., 150.], [5., 5., 60., 90.]]), torch.tensor([[30., 40., 180., 140.]])]
gtc = [torch.tensor([1, 2, 3]), torch.tensor([5])]
rk = torch.randint(0, 2**31-1, (B, 5*7*15), generator=g, dtype=torch.int64)
ok = torch.randint(0, 2**31-1, (B, 2100), generator=g, dtype=torch.int64)
def run(sd, props=None):
    sd = om.clone_state(sd, requires_grad=True)
    l, aux = om.student_losses(sd, imgs, gtb, gtc, list(rk), list(ok), cfg, return_aux=True, proposals=props)
    sum(l.values()).backward()
    return sd, l, aux
sdA, lA, auxA = run(sd0)
# perturb every weight by 1e-6 relative noise (emulates a different fp32 summation order)
sd1 = {k: (v * (1 + EPS * torch.randn(v.shape, generator=g)) if v.dtype == torch.float32 and 'running' not in k else v) for k, v in sd0.items()}
sdB, lB, auxB = run(sd1, props=auxA["props"])
def rel(a, b): return ((a.double()-b.double()).norm()/(b.double().norm()+1e-30)).item()
print("feat rel", rel(auxB["feat"], auxA["feat"]))
for k in ["backbone.vgg0.0.weight", "backbone.vgg2.3.weight", "backbone.vgg4.0.weight", "backbone.vgg4.6.weight", "backbone.vgg4.7.bias", "roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc2.weight", "proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.conv.bias",
          "proposal_generator.rpn_head.anchor_deltas.weight", "roi_heads.box_predictor.cls_score.weight"]:
    print(k, "%.3e" % rel(sdB[k].grad, sdA[k].grad))
