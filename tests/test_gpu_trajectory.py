"""Trajectory parity of the composed step (run_step, source_free_adaptive_teacher.py:335-603).

Three full steps of ``SourceFreeAdaptiveTeacherTrainer`` on the HIP path -- teacher forward (train-mode BatchNorm) ->
score threshold -> student forward / backward on the pseudo labels -> SGD(momentum, weight decay, warm-up LR) -> EMA of
every parameter and buffer -- against the same three steps of the CPU oracle (``om.teacher_forward / student_losses /
sgd_step / ema_update``).  After EVERY step the student's parameters (as updates, i.e. relative to the previous step),
momentum buffers, the teacher's parameters, all 2 x 26 BatchNorm running statistics and the num_batches_tracked
counters are compared.

The oracle is fed the device's discrete decisions (pseudo labels, proposal set; same forced sampling keys), as in
``test_gpu_model._student_vs_oracle``: a 1e-7 difference legitimately flips a rank / NMS decision.  The reference runs
the student's backbone three times per step on the same batch (supervised_target + the zero-weighted domain branch on
k and q, q = k here): the oracle does exactly that; the device runs it once (SFOD.ELIDE_DEAD_BRANCHES) and applies the
two extra momentum updates in closed form -- and, in the second parametrisation, runs everything (ELIDE False).
The EMA rate is lowered (keep 0.9) and the LR set to 2.5e-5 without warm-up: high enough that a step moves the weights
~100x above fp32 resolution, low enough that the planted classifier (x30 weights: logits react to a weight change ~900x
more strongly than at a normal initialisation) keeps producing pseudo labels over the three steps.

Before every step the oracle's state (student, teacher, momentum) is RE-SYNCHRONISED to the device's: the step is
chaotic at the bit level (a ReLU mask, a max-pool arg-max or the sign of an L1 residual flips on 1e-7 noise and moves a
gradient by 1e-3 .. 1e-2, in the reference as much as here -- tests/diagnostics/grad_sensitivity.py), so two independent
fp32 trajectories drift apart step by step no matter how exact each step is; what CAN be pinned is every step of the
device's own trajectory, from the state it actually was in (momentum warm from step 1 on, EMA'd teacher, advanced
running statistics and counters), against the oracle's step from that same state.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda"
HOT_YAML = os.path.join(os.path.dirname(GOLDEN), "..", "configs",
                        "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


R101_YAML = os.path.join(os.path.dirname(GOLDEN), "..", "configs", "r101_c4_cs_foggy_adaptive_teacher_source_free.yaml")


@pytest.mark.parametrize("model,dtype,elide", [("vgg", "fp32", True), ("vgg", "bf16x3", True), ("vgg", "fp32", False),
                                               ("r101", "fp32", True), ("vgg", "f16x3", True), ("r101", "f16x3", True), ("vgg", "f16x3", False)])
def test_three_steps_match_the_oracle_trajectory(sfod, native, model, dtype, elide):
    """``r101``: BASELINE config #5 (r101_c4_cs_foggy_adaptive_teacher_source_free.yaml) in its parity mode -- frozen
    stem / res2 (never move, no momentum), live BatchNorm res3 / res4 refreshed by teacher and student (AdaBN), no
    domain branch in that yaml (DOMAIN_CLASSIFIER defaults: one backbone pass per step, DC parameters untouched);
    WEAK_STRONG_AUGMENT is switched off here so that teacher and student see the captured frames."""
    # Runs under SFOD.DETERMINISTIC: no float atomics in any gradient (the generic weight-gradient kernels sum their pixel
    # splits through slabs in a fixed order), so the device's trajectory is bit-identical from run to run
    # (test_a_step_is_bit_reproducible_under_deterministic_mode) and a case either passes every time or fails every time --
    # the retry this test needed while WHICH near-tie flipped depended on the atomics' arrival order is gone.  The
    # tolerances on the updates stay at the oracle's own flip sensitivity: a ReLU mask / arg-max decision within rounding of
    # a tie falls differently in any two fp32 implementations (tests/diagnostics/grad_sensitivity.py); the arithmetic itself is
    # pinned at 3e-5 / 2e-4 by the flip-free test (tests/test_gpu_flipfree.py).
    _trajectory_case(sfod, native, model, dtype, elide)


def _trajectory_case(sfod, native, model, dtype, elide):
    B, H, W, KEEP, LR, STEPS = 2, 256, 384, 0.9, 2.5e-5, 3
    resnet = model == "r101"
    cfg = sfod.config.setup_cfg(R101_YAML if resnet else HOT_YAML, [
        "OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype, "SOLVER.IMS_PER_BATCH_TARGET", str(B),
        "SFOD.SYNTHETIC.HEIGHT", str(H), "SFOD.SYNTHETIC.WIDTH", str(W), "SFOD.SYNTHETIC.NUM_IMAGES", "8",
        "INPUT.MIN_SIZE_TRAIN", f"({H},)", "INPUT.RANDOM_FLIP", "none", "SOLVER.WARMUP_ITERS", "0",
        "SOLVER.BASE_LR", str(LR), "SFOD.EMA.KEEP_RATE", str(KEEP), "SOLVER.CHECKPOINT_PERIOD", "0",
        "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False", "SFOD.ELIDE_DEAD_BRANCHES", str(elide),
        "SFOD.OVERLAP_TEACHER", "True", "WEAK_STRONG_AUGMENT", "False", "SFOD.DETERMINISTIC", "True"])
    dc_on = bool(cfg.DOMAIN_CLASSIFIER.ENABLED)
    passes = 3 if dc_on else 1            # backbone passes of the reference's student per step (supervised_target + domain branch on k and q)
    torch.manual_seed(5)
    tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
    with torch.no_grad():       # planted labels: pseudo ground truth exists from the first step on
        tr.model.roi_heads.box_predictor.cls_score.weight.mul_(4.0 if resnet else 30.0)
        tr._copy_main_model()
        # Break the student == teacher symmetry of iteration 0: an exact copy predicts exactly the boxes its pseudo labels
        # were decoded from, so the L1 box losses (smooth-L1 with beta 0) sit AT their kink and their gradients
        # sign(pred - target) are the sign of fp32 rounding noise -- in the reference as much as here (measured: 18 %
        # of bbox_pred's gradient).  A student 2 % away from the teacher is what every later iteration looks like.
        gen = torch.Generator(device=DEV).manual_seed(11)
        for n, p in tr.model.named_parameters():
            if not n.startswith("DC_"):
                p.mul_(1.0 + 0.02 * torch.randn(p.shape, device=DEV, generator=gen))
    # the closed-form factor is scoped to run_step's student pass (any other train-mode forward counts once)
    assert tr._elided_bn_updates == (3 if (elide and dc_on) else 1) and tr.model.backbone.bn_updates_per_forward == 1
    ocfg = om.Cfg.r101_c4() if resnet else om.Cfg()
    state = lambda m: om.clone_state({k: v.detach().float().cpu() if v.dtype != torch.int64 else v.detach().cpu().clone()
                                      for k, v in m.state_dict().items()})
    sd_s, sd_t = state(tr.model), state(tr.model_teacher)
    for k, v in sd_s.items():
        if om.is_param(k):
            v.requires_grad_(True)
    bufs = {}
    Hf, Wf = (-(-H // 16), -(-W // 16)) if resnet else (H // 32, W // 32)
    g = torch.Generator().manual_seed(1)
    rpn_keys = torch.randint(0, 2 ** 31 - 1, (B, Hf * Wf * ocfg.num_anchors), generator=g, dtype=torch.int64)
    roi_keys = torch.randint(0, 2 ** 31 - 1, (B, 2100), generator=g, dtype=torch.int64)
    tr.model.proposal_generator._forced_keys = rpn_keys.to(torch.int32).to(DEV)
    tr.model.roi_heads._forced_keys = roi_keys.to(torch.int32).to(DEV)

    cap = {}
    s_rpn = tr.model.proposal_generator
    orig_props, orig_teacher = s_rpn._proposals, tr._teacher_pass

    def cap_props(*a, **k):
        p = orig_props(*a, **k)
        cap.setdefault("props", []).append(p)      # non-elided mode: later calls belong to the dead branches
        return p

    def cap_teacher(data_k):
        cap["images"] = [d["image"].cpu().clone() for d in data_k]
        cap["pseudo"] = orig_teacher(data_k)
        return cap["pseudo"]
    s_rpn._proposals, tr._teacher_pass = cap_props, cap_teacher

    def tol(name, it):
        """every step starts from identical states: the gradient tolerances of
        test_gpu_model.test_student_losses_and_gradients_match_oracle (flip sensitivity of the oracle itself)"""
        x3 = dtype in ("bf16x3", "f16x3")      # f16x3: its backward products are bf16x3's
        if name.startswith("backbone"):
            if resnet:      # 30 live blocks: the flip noise of tests/test_gpu_resnet.py's gradient check
                return 8e-2
            return 6e-2 if x3 else 4e-2
        if name.startswith("roi_heads"):
            if resnet:      # the heads read res4 features that already carry the network's 1e-4 forward noise (DESIGN section 2
                return 8e-3  # of DESIGN.md): more ReLU flips in fc1 / fc2 than behind VGG16 (seen: 2.2e-3)
            return 4e-3 if x3 else 2e-3
        if ".rpn_head." in name:        # few hidden units under sparse gradients: ONE flipped ReLU shows as 1e-3 .. 2e-2
                                        # in the 3x3 conv's gradient and, through the hidden map, in the 1x1 heads' (seen:
                                        # anchor_deltas.weight 1.4e-2 at step 2 in 1 of 3 runs, the others 1e-4);
            return 3e-2                 # (fp32 mode too: its weight gradients use float atomics, so WHICH unit flips varies
                                        # from run to run -- 1 run in 8 reached 1.6e-2 at step 2)
        return 2e-3 if (x3 or resnet) else 2e-4

    names = [n for n, p_ in tr.model.named_parameters() if p_.requires_grad]     # frozen stem / res2: no gradient, no momentum
    frozen = [n for n, p_ in tr.model.named_parameters() if not p_.requires_grad]
    assert bool(frozen) == resnet
    worst, n_pseudo = {}, []
    def resync():
        """oracle state <- device state (parameters, buffers, momentum)"""
        for sd, m in ((sd_s, tr.model), (sd_t, tr.model_teacher)):
            for k, v in m.state_dict().items():
                with torch.no_grad():
                    sd[k].copy_(v.detach().cpu())
        if tr.optimizer._steps > 0:      # momentum buffers exist from the first optimiser step on
            for n in names:
                o, k, shp = tr.optimizer.flat.offsets[n]
                bufs[n] = tr.optimizer.mom[o:o + k].view(shp).detach().cpu().clone()

    for it in range(STEPS):
        if it > 0:
            resync()
        prev_dev = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
        prev_ref = {n: sd_s[n].detach().clone() for n in names}
        prev_t_dev = {n: p.detach().clone() for n, p in tr.model_teacher.named_parameters()}
        prev_t_ref = {n: sd_t[n].detach().clone() for n in names}
        cap.pop("props", None)
        tr.iter = it
        tr.run_step()
        tr.after_step()
        rec = tr.storage.flush()          # THIS step's values (the periodic writer reports d2's median over the last 20 steps)
        tr._flush_metrics()               # ... and the writer's own path: finiteness checks, nothing pending
        torch.cuda.synchronize()
        # ---- the same step on the oracle ---------------------------------------------------------------------------
        images = cap["images"]
        om.teacher_forward(sd_t, images, ocfg)                    # refreshes the teacher's running statistics (q2)
        ps = cap["pseudo"]
        gtb = [ps.boxes[b, : ps.count[b].item()].cpu() for b in range(B)]
        gtc = [ps.classes[b, : ps.count[b].item()].cpu().long() for b in range(B)]
        n_pseudo.append(sum(len(x) for x in gtb))
        assert it > 0 or n_pseudo[0] > 0      # later steps may legitimately have none (images without ground truth)
        pr = cap["props"][0]
        given = [(pr.boxes[b, : pr.count[b].item()].cpu(), pr.logits[b, : pr.count[b].item()].cpu()) for b in range(B)]
        for v in sd_s.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        losses = om.student_losses(sd_s, images, gtb, gtc, list(rpn_keys), list(roi_keys), ocfg, proposals=given)
        with torch.no_grad():       # the reference's two further backbone passes of the domain branch (same batch, q = k)
            x, _ = om.preprocess(images)
            for _ in range(passes - 1):
                om.backbone_forward(sd_s, x, ocfg, training=True)
        sum(v for k, v in losses.items() if k != "loss_bpc").backward()
        grads = {}
        for n in names:     # zero-weighted branches still hand autograd a (zero) gradient: weight decay applies;
            gr = sd_s[n].grad      # a domain classifier that never runs leaves None: SGD skips the parameter
            grads[n] = gr if gr is not None else (torch.zeros_like(sd_s[n]) if dc_on or not n.startswith("DC_") else None)
        om.sgd_step(sd_s, grads, bufs, lr=om.lr_at(it, LR, warmup_iters=0))
        om.ema_update(sd_t, {k: v.detach() for k, v in sd_s.items()}, KEEP)
        # ---- compare ---------------------------------------------------------------------------------------------------
        for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg"):
            np.testing.assert_allclose(rec[k + "_pseudo"], losses[k].item(), rtol=1e-4, atol=1e-7, err_msg=f"step {it} {k}")
        mom = tr.optimizer.mom
        flat = tr.optimizer.flat
        for n, p in tr.model.named_parameters():
            if n in frozen:     # FREEZE_AT 2: bit-identical to where it started
                assert torch.equal(p.detach(), prev_dev[n]), (it, n)
                continue
            if n.startswith("DC_"):
                # zero gradient + weight decay: the parameter shrinks by lr * (wd * p [+ momentum]) exactly as in the oracle
                # (domain classifier off, r101 yaml: no gradient at all, the parameter does not move)
                torch.testing.assert_close(p.detach().cpu(), sd_s[n].detach(), rtol=1e-6, atol=1e-9)
                if not dc_on:
                    assert torch.equal(p.detach(), prev_dev[n]), (it, n)
                continue
            parts = n.split(".")
            if parts[0] == "backbone" and parts[1].startswith("vgg") and parts[-1] == "bias" and parts[2] in ("0", "3", "6"):
                continue        # conv bias in front of train-mode BatchNorm: analytically zero gradient (deviation 4)
            d_dev = (p.detach() - prev_dev[n]).cpu()
            d_ref = sd_s[n].detach() - prev_ref[n]
            e = rel_err(d_dev, d_ref)
            worst[(it, n)] = e
            # the update is read back as a difference of two fp32 parameter tensors: allow for their rounding
            ulp = 2 * 6e-8 * sd_s[n].detach().double().norm().item() / (d_ref.double().norm().item() + 1e-30)
            assert e < tol(n, it) + ulp, (it, n, e, ulp)
            o, k, shp = flat.offsets[n]
            e_m = rel_err(mom[o:o + k].view(shp), bufs[n])
            assert e_m < tol(n, it), (it, n, "momentum", e_m)
        t_sd = tr.model_teacher.state_dict()
        s_sd = tr.model.state_dict()
        for n, v in t_sd.items():
            if v.dtype == torch.int64:
                assert int(v) == int(sd_t[n]), (it, n, int(v), int(sd_t[n]))
                assert int(s_sd[n]) == int(sd_s[n]) == passes * (it + 1), (it, n)
            elif "running" in n:
                # (ResNet: the reference arithmetic's own forward noise is 20x VGG's, tests/test_gpu_fullsize.py header)
                a, r = (1e-6 if (dtype == "fp32" and not resnet) else 2e-5), 2e-4
                torch.testing.assert_close(v.cpu(), sd_t[n].detach(), rtol=r, atol=a, msg=lambda m: f"teacher {n} step {it}: {m}")
                torch.testing.assert_close(s_sd[n].cpu(), sd_s[n].detach(), rtol=r, atol=a, msg=lambda m: f"student {n} step {it}: {m}")
            elif n in names and not n.startswith("DC_"):
                # teacher = EMA of the student: its update is (1 - keep) * (student - teacher)
                parts = n.split(".")
                if not (parts[0] == "backbone" and parts[1].startswith("vgg") and parts[-1] == "bias" and parts[2] in ("0", "3", "6")):
                    d_t = sd_t[n].detach() - prev_t_ref[n]
                    e_t = rel_err(v.detach() - prev_t_dev[n], d_t)
                    ulp = 2 * 6e-8 * sd_t[n].detach().double().norm().item() / (d_t.double().norm().item() + 1e-30)
                    assert e_t < tol(n, it) + 1e-3 + ulp, (it, n, "teacher update", e_t, ulp)
    s_rpn._proposals, tr._teacher_pass = orig_props, orig_teacher
    native.set_deterministic(False)
    print(f"\n[trajectory {model} {dtype} elide={elide}] pseudo labels per step {n_pseudo}; worst update error per group after {STEPS} steps: " + ", ".join(
        f"{grp} {max(v for (i, n), v in worst.items() if n.startswith(grp)):.2e}"
        for grp in ("backbone", "proposal_generator", "roi_heads")))


@pytest.mark.parametrize("model,dtype", [("vgg", "bf16x3"), ("vgg", "fp32"), ("r101", "f16x3")])
def test_a_step_is_bit_reproducible_under_deterministic_mode(sfod, native, model, dtype):
    """SFOD.DETERMINISTIC: two trainers from the same seed, three teacher+student steps each (teacher on the second stream)
    -> every parameter, momentum buffer, running statistic and logged loss is bit-identical.  Without the mode the generic
    weight gradients (1x1 / linear / first layer) combine their pixel splits with float atomics and the second run differs
    in the last bits (which this test also records: the default mode must differ somewhere, or the switch switches
    nothing)."""
    B, H, W = 2, 256, 384
    resnet = model == "r101"

    def run(det):
        cfg = sfod.config.setup_cfg(R101_YAML if resnet else HOT_YAML, [
            "OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype, "SOLVER.IMS_PER_BATCH_TARGET", str(B),
            "SFOD.SYNTHETIC.HEIGHT", str(H), "SFOD.SYNTHETIC.WIDTH", str(W), "SFOD.SYNTHETIC.NUM_IMAGES", "8",
            "INPUT.MIN_SIZE_TRAIN", f"({H},)", "SOLVER.WARMUP_ITERS", "0", "SOLVER.BASE_LR", "2.5e-5",
            "SOLVER.CHECKPOINT_PERIOD", "0", "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False", "SFOD.EVAL_HOOK", "False",
            "WEAK_STRONG_AUGMENT", "False", "SFOD.DETERMINISTIC", str(det)])
        torch.manual_seed(5)
        tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
        with torch.no_grad():
            tr.model.roi_heads.box_predictor.cls_score.weight.mul_(4.0 if resnet else 30.0)
            tr._copy_main_model()
        assert bool(native.load().sfod_get_deterministic()) == det
        g = torch.Generator().manual_seed(1)
        A = 12 if resnet else 15
        Hf, Wf = (-(-H // 16), -(-W // 16)) if resnet else (H // 32, W // 32)
        tr.model.proposal_generator._forced_keys = torch.randint(0, 2 ** 31 - 1, (B, Hf * Wf * A), generator=g).to(torch.int32).to(DEV)
        tr.model.roi_heads._forced_keys = torch.randint(0, 2 ** 31 - 1, (B, 2100), generator=g).to(torch.int32).to(DEV)
        losses = []
        for it in range(3):
            tr.iter = it
            tr.run_step()
            tr.scheduler.step()
            rec = tr.storage.flush()
            losses.append([rec[k] for k in sorted(rec) if k.startswith("loss")])
        torch.cuda.synchronize()
        f, tf = tr.optimizer.flat, tr.teacher_flat
        out = [f.param.clone(), tr.optimizer.mom.clone(), f.fbuf.clone(), tf.param.clone(), tf.fbuf.clone()], losses
        del tr
        torch.cuda.empty_cache()
        return out

    try:
        (a, la), (b, lb) = run(True), run(True)
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"deterministic mode: state tensor {i} differs between two runs"
        assert la == lb, "deterministic mode: logged losses differ between two runs"
        (c, _), (d, _) = run(False), run(False)
        same = all(torch.equal(x, y) for x, y in zip(c, d))
        print(f"\n[deterministic {model} {dtype}] two deterministic runs: bit-identical; two default runs: "
              f"{'bit-identical too (no split was combined by atomics at this size)' if same else 'differ in the last bits'}")
        # the two modes compute the same sums in another order: after three steps the parameters agree to the level at which
        # two runs of the default mode agree with each other (a last-bit difference in a gradient flips ReLU / arg-max ties
        # in the following steps: 1e-7 ... 1e-4 relative, by network)
        d_modes = ((a[0] - c[0]).norm() / (c[0].norm() + 1e-30)).item()
        d_runs = ((c[0] - d[0]).norm() / (c[0].norm() + 1e-30)).item()
        print(f"[deterministic {model} {dtype}] parameters after 3 steps: deterministic vs default {d_modes:.1e}, default vs default {d_runs:.1e}")
        assert d_modes < max(1e-5, 10 * d_runs)
    finally:
        native.set_deterministic(False)


def test_bf16x3_stays_with_fp32_over_a_longer_horizon(sfod, native):
    """Free-running comparison (no re-synchronisation): 16 teacher+student steps of the hot yaml from the same seed in
    ``fp32``, in ``fp32`` from weights perturbed by 1e-7 relative (= "another correct fp32 implementation": the step is
    chaotic at the bit level -- a detection crossing the 0.8 threshold, an NMS or ReLU flip -- so even that run leaves
    the first one), and in ``bf16x3``.  What is pinned is the SCALE of the split-precision mode's drift: after 16 steps
    its student is not further from the fp32 student than a few times the distance between the two fp32 runs, and both
    stay a small fraction of the distance the weights travelled."""
    B, H, W, STEPS = 2, 256, 384, 16
    runs = {}
    for tag, dtype, eps in (("fp32", "fp32", 0.0), ("fp32_perturbed", "fp32", 1e-7), ("bf16x3", "bf16x3", 0.0), ("f16x3", "f16x3", 0.0)):
        cfg = sfod.config.setup_cfg(HOT_YAML, [
            "OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype, "SOLVER.IMS_PER_BATCH_TARGET", str(B),
            "SFOD.SYNTHETIC.HEIGHT", str(H), "SFOD.SYNTHETIC.WIDTH", str(W), "SFOD.SYNTHETIC.NUM_IMAGES", "8",
            "INPUT.MIN_SIZE_TRAIN", f"({H},)", "SOLVER.WARMUP_ITERS", "0", "SOLVER.BASE_LR", "2.5e-4",
            "SOLVER.CHECKPOINT_PERIOD", "0", "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False", "SFOD.EVAL_HOOK", "False"])
        torch.manual_seed(11)
        tr = sfod.engine.SourceFreeAdaptiveTeacherTrainer(cfg)
        with torch.no_grad():
            tr.model.roi_heads.box_predictor.cls_score.weight.mul_(30.0)
            if eps:
                gen = torch.Generator(device=DEV).manual_seed(3)
                f = tr.optimizer.flat.param
                f.mul_(1.0 + eps * torch.randn(f.shape, device=DEV, generator=gen))
            tr._copy_main_model()
        g = torch.Generator().manual_seed(1)      # identical sampling keys in all runs
        tr.model.proposal_generator._forced_keys = torch.randint(0, 2 ** 31 - 1, (B, (H // 32) * (W // 32) * 15),
                                                                 generator=g).to(torch.int32).to(DEV)
        tr.model.roi_heads._forced_keys = torch.randint(0, 2 ** 31 - 1, (B, 2100), generator=g).to(torch.int32).to(DEV)
        p0 = tr.optimizer.flat.param.clone()
        hist = []
        for it in range(STEPS):
            tr.iter = it
            tr.run_step()
            tr.scheduler.step()
            rec = tr.storage.flush()
            hist.append([rec[k] for k in ("loss_rpn_cls_pseudo", "loss_rpn_loc_pseudo", "loss_cls_pseudo", "loss_box_reg_pseudo")])
        runs[tag] = (np.array(hist), tr.optimizer.flat.param.clone(), p0)
        del tr
        torch.cuda.empty_cache()
    h32, p32, p0 = runs["fp32"]
    assert all(np.isfinite(runs[k][0]).all() for k in runs)
    moved = (p32 - p0).norm().item()
    d_fp32 = (runs["fp32_perturbed"][1] - p32).norm().item()
    d_x3 = (runs["bf16x3"][1] - p32).norm().item()
    med = {k: np.median(np.abs(runs[k][0] - h32) / np.maximum(np.abs(h32), 1e-3), axis=0) for k in ("fp32_perturbed", "bf16x3")}
    print(f"\n[{STEPS} free-running steps] student distance from the fp32 run / distance travelled: another fp32 run "
          f"{d_fp32 / moved:.3f}, bf16x3 {d_x3 / moved:.3f}; median relative loss difference per key: fp32' {med['fp32_perturbed']}, "
          f"bf16x3 {med['bf16x3']}")
    d_h3 = (runs["f16x3"][1] - p32).norm().item()
    print(f"[longer horizon] f16x3 distance / moved {d_h3 / moved:.3f}")
    # all three distances are outcomes of the same chaotic process (seen: 0.34 ... 0.41 of the distance travelled, any of
    # them the largest): the modes may not drift more than twice as far as the second fp32 run does, nor leave the
    # neighbourhood of the fp32 trajectory altogether
    for d in (d_x3, d_h3):
        assert d < 2.0 * d_fp32 + 0.05 * moved, (d / moved, d_fp32 / moved)
        assert d < 0.9 * moved, d / moved
