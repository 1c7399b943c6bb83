import os, sys, importlib, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import model as om
sfod = importlib.import_module("simple-sfod_amd")
HOT = "/root/repo/configs/faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"
B, H, W, KEEP, LR = 2, 160, 224, 0.9, 0.0025
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = sfod.config.setup_cfg(HOT, ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype, "SOLVER.IMS_PER_BATCH_TARGET", str(B),
    "SFOD.SYNTHETIC.HEIGHT", str(H), "SFOD.SYNTHETIC.WIDTH", str(W), "SFOD.SYNTHETIC.NUM_IMAGES", "8",
    "INPUT.MIN_SIZE_TRAIN", f"({H},)", "INPUT.RANDOM_FLIP", "none", "SOLVER.WARMUP_ITERS", "0",
    "SOLVER.BASE_LR", str(LR), "SFOD.EMA.KEEP_RATE", str(KEEP), "SOLVER.CHECKPOINT_PERIOD", "0",
    "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False"])
torch.manual_seed(5)
tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
with torch.no_grad():
    tr.model.roi_heads.box_predictor.cls_score.weight.mul_(60.0); tr._copy_main_model()
state = lambda m: om.clone_state({k: v.detach().float().cpu() if v.dtype != torch.int64 else v.detach().cpu().clone() for k, v in m.state_dict().items()})
sd_s, sd_t = state(tr.model), state(tr.model_teacher)
for k, v in sd_s.items():
    if om.is_param(k): v.requires_grad_(True)
Hf, Wf = H // 32, W // 32
g = torch.Generator().manual_seed(1)
rk = torch.randint(0, 2**31-1, (B, Hf*Wf*15), generator=g, dtype=torch.int64)
ok = torch.randint(0, 2**31-1, (B, 2100), generator=g, dtype=torch.int64)
tr.model.proposal_generator._forced_keys = rk.to(torch.int32).cuda(); tr.model.roi_heads._forced_keys = ok.to(torch.int32).cuda()
cap = {}
rpn = tr.model.proposal_generator
op, ot = rpn._proposals, tr._teacher_pass
def cp(*a, **k):
    p = op(*a, **k); cap.setdefault("props", []).append(p); return p
def ct(d):
    cap["images"] = [x["image"].cpu().clone() for x in d]; cap["pseudo"] = ot(d); return cap["pseudo"]
rpn._proposals, tr._teacher_pass = cp, ct
# run the step but stop before the optimizer: replicate run_step pieces
orig_step = tr.optimizer.step
tr.optimizer.step = lambda **k: None
tr.iter = 0
tr.run_step()
torch.cuda.synchronize()
flat = tr.optimizer.flat
ps = cap["pseudo"]
gtb = [ps.boxes[b, :ps.count[b].item()].cpu() for b in range(B)]
gtc = [ps.classes[b, :ps.count[b].item()].cpu().long() for b in range(B)]
print("pseudo counts", [len(x) for x in gtb])
pr = cap["props"][0]
given = [(pr.boxes[b, :pr.count[b].item()].cpu(), pr.logits[b, :pr.count[b].item()].cpu()) for b in range(B)]
losses = om.student_losses(sd_s, cap["images"], gtb, gtc, list(rk), list(ok), om.Cfg(), proposals=given)
sum(v for k, v in losses.items() if k != "loss_bpc").backward()
def rel(a, b): return ((a.double()-b.double()).norm()/(b.double().norm()+1e-30)).item()
for n, p in tr.model.named_parameters():
    if n.startswith("DC_") : continue
    o, k, shp = flat.offsets[n]
    gd = flat.grad[o:o+k].view(shp).cpu()
    gr = sd_s[n].grad
    if gr is None: print(n, "no ref grad"); continue
    e = rel(gd, gr)
    if e > 1e-3 or "rpn" in n or "roi" in n: print(f"{n:55s} {e:.3e}  |g| {gr.norm():.3e} |p| {sd_s[n].detach().norm():.3e}")
