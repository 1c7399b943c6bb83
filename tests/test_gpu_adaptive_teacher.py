"""``TRAINER: "adaptive_teacher"`` on the device (daod/engine/trainers/adaptive_teacher.py:191-357,
daod/modeling/meta_arch/adaptive_teacher_rcnn.py:210-257): the with-source teacher-student loop on the HIP kernels.

  * the ``supervised`` branch's five losses against the CPU oracle on identical weights / frames / sampling keys (the four
    detection losses through ``oracle.model.student_losses``; ``loss_DC_img_s`` = 0.001 x BCE of the image-level
    discriminator on the oracle's ``vgg4`` features, label 0, through the gradient-reversal layer), and the gradient the
    reversed discriminator loss sends into the backbone;
  * the schedule on the real trainer: burn-in steps leave the teacher alone, at ``BURN_UP_STEP`` the teacher becomes the
    student bit for bit (parameters, running statistics, counters), afterwards every ``TEACHER_UPDATE_ITER`` steps
    ``student * (1 - k) + teacher * k`` with the reference's operation order, bit for bit, and nothing in between;
  * after burn-in all three student branches contribute: the logged ``loss_DC_img_s`` is the domain pass's (~ln 2), not the
    ``supervised`` branch's x 0.001; the discriminator's weights move.
The orchestration itself (call order, lists, weights) is pinned on the CPU against the reference's own ``run_step``:
tests/test_oracle_adaptive_teacher.py.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import model as om
from test_gpu_model import make_inputs, oracle_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
YAML = os.path.join(os.path.dirname(GOLDEN), "..", "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher.yaml")
SMALL = ["SFOD.SYNTHETIC.HEIGHT", "256", "SFOD.SYNTHETIC.WIDTH", "512", "SFOD.SYNTHETIC.NUM_IMAGES", "4",
         "INPUT.MIN_SIZE_TRAIN", "(192,)", "SOLVER.CHECKPOINT_PERIOD", "0", "TEST.EVAL_PERIOD", "0", "TEST.VAL_LOSS", "False"]


def _dc_img_torch(sd, feat):
    """FCDiscriminator_img (daod/modeling/dann/dann.py:10-29) in plain torch on the oracle's features"""
    F = torch.nn.functional
    h = feat
    for name in ("conv1", "conv2", "conv3"):
        h = F.leaky_relu(F.conv2d(h, sd[f"DC_img.{name}.weight"], sd[f"DC_img.{name}.bias"], padding=1), 0.2)
    return F.conv2d(h, sd["DC_img.classifier.weight"], sd["DC_img.classifier.bias"], padding=1)


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_supervised_branch_losses_and_reversed_discriminator_gradient_match_the_oracle(sfod, native, dtype):
    cfg = sfod.config.setup_cfg(YAML, ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype])
    torch.manual_seed(3)
    model = sfod.modeling.build_model(cfg).train()
    assert type(model).__name__ == "AdaptiveTeacherGeneralizedRCNN"
    B, H, W = 2, 192, 384
    inputs = make_inputs(B, H, W, [5, 3], seed=11)
    sd = oracle_state(model)
    ocfg = om.Cfg()
    Hf, Wf = H // 32, W // 32
    g = torch.Generator().manual_seed(12)
    rpn_keys = torch.randint(0, 2 ** 31 - 1, (B, Hf * Wf * 15), generator=g, dtype=torch.int64)
    roi_keys = torch.randint(0, 2 ** 31 - 1, (B, 2100), generator=g, dtype=torch.int64)
    model.proposal_generator._forced_keys = rpn_keys.to(torch.int32).to(DEV)
    model.roi_heads._forced_keys = roi_keys.to(torch.int32).to(DEV)
    cap = {}
    orig = model.proposal_generator._proposals
    model.proposal_generator._proposals = lambda *a, **k: cap.setdefault("p", orig(*a, **k))
    losses, a, b = model(inputs, branch="supervised")
    model.proposal_generator._proposals = orig
    assert a == [] and b == [] and sorted(losses) == ["loss_DC_img_s", "loss_box_reg", "loss_cls", "loss_rpn_cls", "loss_rpn_loc"]
    # only the discriminator term goes backward here: what arrives in the backbone is the REVERSED gradient
    for p in model.parameters():
        p.grad = None
    losses["loss_DC_img_s"].backward()
    g_dc = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    pr = cap["p"]
    given = [(pr.boxes[i, : pr.count[i].item()].cpu(), pr.logits[i, : pr.count[i].item()].cpu()) for i in range(B)]
    ref = om.student_losses(sd, [d["image"] for d in inputs], [d["instances"].gt_boxes.tensor for d in inputs],
                            [d["instances"].gt_classes for d in inputs], list(rpn_keys), list(roi_keys), ocfg, proposals=given)
    tol = 1e-4 if dtype == "fp32" else 2e-4
    for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"):
        np.testing.assert_allclose(losses[k].item(), ref[k].item(), rtol=tol, err_msg=k)
    # discriminator: GRL(-1) -> 4 convs -> BCE-with-logits(label 0) * 0.001 on the oracle's features of the same frames
    x, _ = om.preprocess([d["image"] for d in inputs])
    sd2 = oracle_state(model)            # (student_losses moved sd's running statistics; train-mode BN does not read them)
    feat = om.backbone_forward(sd2, x, ocfg, training=True)
    logits = _dc_img_torch(sd2, feat)
    l_ref = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.zeros_like(logits)) * 0.001
    np.testing.assert_allclose(losses["loss_DC_img_s"].item(), l_ref.item(), rtol=tol)
    # gradients: the discriminator's own parameters get d(loss)/dw; the backbone gets MINUS d(loss)/d(feat) chained down
    gl = torch.autograd.grad(l_ref, [sd2["DC_img.conv1.weight"], sd2["DC_img.classifier.bias"], feat], retain_graph=True)
    rel = lambda u, v: ((u.double() - v.double()).norm() / (v.double().norm() + 1e-30)).item()
    gt = 2e-4 if dtype == "fp32" else 2e-3
    assert rel(g_dc["DC_img.conv1.weight"], gl[0]) < gt and rel(g_dc["DC_img.classifier.bias"], gl[1]) < gt
    (-feat * gl[2].detach()).sum().backward()          # the reversed gradient, chained through the oracle's backbone
    name = "backbone.vgg4.0.weight"
    # (a weight gradient in front of a train-mode BatchNorm over 2 x 6 x 12 pixels is a difference of nearly equal sums:
    # measured 4.7e-3 in fp32 at this size; the sign and direction are what this assertion is about)
    ref_g = sd2[name].grad
    cos = (g_dc[name].double().flatten() @ ref_g.double().flatten()) / (g_dc[name].double().norm() * ref_g.double().norm())
    assert name in g_dc and rel(g_dc[name], ref_g) < (2e-2 if dtype == "fp32" else 5e-2) and cos.item() > 0.999
    del model.proposal_generator._forced_keys, model.roi_heads._forced_keys


def test_burn_in_hand_over_and_ema_schedule_on_the_device(sfod, native):
    BURN, EVERY, KEEP = 2, 2, 0.9
    cfg = sfod.config.setup_cfg(YAML, ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", "fp32", "SEMISUPNET.BURN_UP_STEP", str(BURN),
                                       "SEMISUPNET.TEACHER_UPDATE_ITER", str(EVERY), "SEMISUPNET.EMA_KEEP_RATE", str(KEEP),
                                       "SEMISUPNET.DIS_LOSS_WEIGHT", "0.1", "SOLVER.MAX_ITER", "8", "SOLVER.BASE_LR", "0.01"] + SMALL)
    torch.manual_seed(cfg.SEED)
    tr = sfod.engine.AdaptiveTeacherTrainer(cfg)
    assert type(tr.model).__name__ == "AdaptiveTeacherGeneralizedRCNN" and tr.ema_enabled is False
    assert all(p.requires_grad for n, p in tr.model.named_parameters() if n.startswith("DC_img."))
    # planted head on the teacher's side of things: after the hand-over some detections must clear 0.8
    with torch.no_grad():
        bp = tr.model.roi_heads.box_predictor
        bp.cls_score.weight.mul_(60.0)

    def state(m):
        return {k: v.detach().clone() for k, v in m.state_dict().items()}
    t_init = state(tr.model_teacher)
    dc0 = tr.model.DC_img.conv1.weight.detach().clone()
    recs = {}
    prev_student = None
    for it in range(7):
        tr.iter = it
        if it == BURN:      # a head that labels something (engine/planted.py: background bias bisected on target frames)
            frames = next(tr._data_loader_iter)[3]
            planted = sfod.engine.planted.plant_model(tr.model, frames, 1.0, target=6.0)
            print(f"\n[adaptive_teacher] planted head at the hand-over: {planted}")
        before_t, before_s = state(tr.model_teacher), state(tr.model)
        tr.run_step()
        tr.scheduler.step()
        torch.cuda.synchronize()
        recs[it] = tr.storage.flush()
        after_t = state(tr.model_teacher)
        float_keys = [k for k, v in before_t.items() if v.is_floating_point() and "running" not in k]
        if it < BURN:
            # burn-in: the teacher is not touched at all (not even run: its BatchNorm statistics stay)
            assert all(torch.equal(after_t[k], t_init[k]) for k in after_t), it
            assert sorted(k for k in recs[it] if k.startswith("loss")) == \
                ["loss_DC_img_s", "loss_box_reg", "loss_cls", "loss_rpn_cls", "loss_rpn_loc"]
            assert 0.2e-3 < recs[it]["loss_DC_img_s"] < 2e-3                      # 0.001 x ~ln 2
        else:
            if it == BURN:
                expect = {k: before_s[k] for k in float_keys}                    # keep_rate 0: the student, bit for bit
            elif (it - BURN) % EVERY == 0:
                # adaptive_teacher.py:349-352: student * (1 - keep_rate) + teacher * keep_rate, fp32, this operation order
                expect = {k: before_s[k] * (1 - KEEP) + before_t[k] * KEEP for k in float_keys}
            else:
                expect = {k: before_t[k] for k in float_keys}                    # no update this step
            for k in float_keys:
                assert torch.equal(after_t[k], expect[k]), (it, k)
            # the teacher ran in train mode afterwards: its running statistics moved on from the (updated) values
            assert any(not torch.equal(after_t[k], before_t[k]) for k in after_t if "running_mean" in k)
            keys = sorted(k for k in recs[it] if k.startswith("loss"))
            assert keys == sorted(["loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_DC_img_s", "loss_DC_img_t"] +
                                  [k + "_pseudo" for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")]), keys
            assert 0.3 < recs[it]["loss_DC_img_s"] < 1.5 and 0.3 < recs[it]["loss_DC_img_t"] < 1.5      # the domain pass's, unweighted
            assert "roi_head/num_pseudo_proposals" in recs[it] and "roi_head/mean_confidence" not in recs[it]
        for k, v in recs[it].items():
            assert np.isfinite(v), (it, k, v)
        prev_student = before_s
    assert max(recs[it]["roi_head/num_pseudo_proposals"] for it in range(BURN, 7)) > 0      # the planted head labels something
    assert not torch.equal(dc0, tr.model.DC_img.conv1.weight.detach())
    assert prev_student is not None
