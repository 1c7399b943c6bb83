"""CPU restatement ("oracle") of the teacher-student Faster R-CNN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``simple-sfod_amd/``) may
import, call, link or execute anything in this directory; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only
as the checker / the timed CPU baseline.

What it restates
----------------
The reference (EPFL-IMOS/simple-SFOD) is a thin layer of Detectron2 subclasses; the
arithmetic of the hot path lives in Detectron2 / torchvision / torch, none of which
except torch is installed here.  So this oracle restates

* the reference's own glue (``daod/modeling/meta_arch/vgg.py``,
  ``daod/modeling/proposal_generator/rpn.py``,
  ``daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py``,
  ``daod/modeling/meta_arch/source_free_adaptive_teacher_rcnn.py``,
  ``daod/engine/trainers/source_free_adaptive_teacher.py``), and
* the upstream semantics it inherits (detectron2 ``main``
  (un-pinned, README says 0.6), torchvision ``nms`` / ``roi_align`` / ``batched_nms``),

in plain fp32 PyTorch-CPU + numpy (+ one small C file for ROIAlign).

Parity pin status
-----------------
* PINNED against the reference itself -- its own code objects, loaded from /root/reference in the build container and
  RUN by ``oracle/gen_golden.py``; only the recorded inputs / outputs are committed (``tests/golden/``):

    fixture            reference code that produced it                                          rows
    vgg_ref.npz        daod/modeling/meta_arch/vgg.py: forward, train-mode BN statistics,       a1 (fwd + bwd)
                       input gradient, PARAMETER gradients of the backward (round 5)
    dann_ref.npz       daod/modeling/dann/dann.py: discriminators, gradient reversal            a12
    adaptive_ref.npz   daod/modeling/adaptive_thresh/adaptive_confidence.py                     f4
    bpc_ref.npz        daod/loss/bpc_loss.py                                                    f4
    glue_ref.npz       (round 5, behind the import hook oracle/ref_stub/hook.py)
                       daod/modeling/roi_heads/source_free_fast_rcnn.py:38-147                  a6
                       daod/engine/trainers/source_free_adaptive_teacher.py:150-183,256-280     a7
                       ... :583-603 (_update_teacher_model incl. int64 counters, DDP prefix)    a9
                       daod/data/common.py:199-228 (aspect bucketing)                           a13
                       daod/modeling/proposal_generator/rpn.py:16-58 (layout, 2nd loss weight)  a4 (glue only)
                       daod/engine/trainers/base.py:318-328 (reset_bn_stats)                    a11
                       daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:68-215 a5 (glue: label_and_sample_proposals on the
                         oracle's d2 primitives, forward / _forward_box control flow)
                       daod/engine/trainers/source_free_adaptive_teacher.py:335-581 (run_step   a8
                         on recorder models: call order, label handling, logged scalars, the
                         weight of every loss key = the gradient backward() leaves on its leaf)
                       daod/engine/trainers/base.py:93-123,186-220 (BaseTrainer.run_step,       a10, e (metrics)
                         _write_metrics with two ranks' dicts: max / mean / total_loss rule)
                       daod/modeling/meta_arch/source_free_adaptive_teacher_rcnn.py:106-339     a3
                         (forward on recorder sub-modules: calls + flags, arity, loss keys per branch)
                       daod/data/build.py:312-367 (per-rank batch for world 1/2/4, the           a13, e
                         divisibility assertion's text, the ASPECT_RATIO_GROUPING error)
    config_ref.json    daod/config.py:8-142 (add_config on a recording node)                    b
    adaptive_teacher_ref.npz   (round 6, the with-source path: SURVEY 8f rank 4's last entry)
                       daod/engine/trainers/adaptive_teacher.py:191-336 (run_step at five        f4 (with-source trainer)
                         iterations around BURN_UP_STEP: branches, lists, EMA schedule, weights,
                         the doubly defined loss_DC_img_s, unweighted metrics)
                       ... :338-357 (_update_teacher_model: keep_rate 0 = copy, EMA, int64)
                       daod/modeling/meta_arch/adaptive_teacher_rcnn.py:102-292 (forward on
                         recorder sub-modules: DC_img before the RPN, x 0.001, three-tuples)
                       daod/data/common.py:119-160 (four-way batch iterator incl. dropped elements)

  ``tests/test_oracle_golden.py`` / ``tests/test_oracle_glue.py`` / ``tests/test_oracle_adaptive_teacher.py`` hold the restatements in this directory to those vectors
  (bit-exact for everything discrete and for the EMA; 1e-4 for the VGG trunk), ``tests/test_gpu_glue.py`` /
  ``tests/test_gpu_model.py`` hold the HIP path to the same vectors directly.
* PINNED against Detectron2's own unit tests (round 5) -- the half that lives in Detectron2: three of its unit tests are
  fully specified by a torch seed, a construction order and literal inputs, and assert literal outputs
  (tests/modeling/test_rpn.py::test_rpn, tests/modeling/test_roi_heads.py::test_roi_heads,
  tests/modeling/test_fast_rcnn.py::test_fast_rcnn).  ``tests/helpers/d2_published.py`` rebuilds their weights and inputs
  (torch's CPU generator stream has not changed; the ResNet-50-C4 backbone they construct is only DRAWN, layer shape by layer
  shape) and quotes the expected values; the oracle reproduces them to 7 digits (tests/test_oracle_d2_golden.py):

    Detectron2 test    expected values                                                           what it pins here
    test_rpn           loss_rpn_cls 0.0804563984, loss_rpn_loc 0.0990132466, 2 + 5 proposal      DefaultAnchorGenerator, Matcher + low-quality
                       boxes and objectness logits after clip + NMS(0.7)                         matches, Box2BoxTransform (1,1,1,1), RPN losses and
                                                                                                 normaliser, find_top_rpn_proposals (decode, clip,
                                                                                                 non-empty, NMS, top-k)          rows a4
    test_roi_heads     loss_cls 4.4236516953, loss_box_reg 0.0091214813 (80 classes, 14 x 14)     ROIAlignV2, proposal_append_gt +
                                                                                                 label_and_sample_proposals, box head,
                                                                                                 FastRCNNOutputLayers.losses      rows a5
    test_fast_rcnn     loss_cls 1.7951188087, loss_box_reg 4.0357131958                          Box2BoxTransform (10,10,5,5), smooth-L1 beta 0
    test_anchor_generator / test_matcher / test_boxes (pairwise_iou) / layers/test_roi_align / test_scheduler:
                       the literal vectors of those tests (tests/test_oracle_ops.py, test_oracle_d2_golden.py)

  ``tests/test_gpu_d2_golden.py`` runs the PRODUCT's ``RPN`` / ``StandardROIHeads`` modules on the HIP kernels against the
  same numbers directly, in all three arithmetic modes.
* PINNED against an independent implementation that IS in this image (round 6): greedy NMS and class-wise ``batched_nms`` --
  HuggingFace ``transformers``' OwlViT post-processing carries its own greedy NMS (argsort by score, fp32 inter / union,
  strict '>': transformers/models/owlvit/image_processing_pil_owlvit.py); ``tests/helpers/hf_nms.py`` CALLS it, and
  ``oracle.box_ops.nms`` returns its keep list (order included) on 64 .. 600 random boxes at three thresholds and on an exact
  IoU == threshold case; ``batched_nms`` -- both the coordinate-offset form and the per-class form above torchvision's 20 000-
  element switch -- returns what that NMS gives when run once per class (tests/test_oracle_hf_nms.py); the HIP kernels do the
  same on 600 / 2000 boxes (tests/test_gpu_ops.py::test_nms_kernel_equals_the_independent_huggingface_implementation).
  ``model.fast_rcnn_inference`` END TO END equals a composition written around that NMS (softmax -> per-class decode -> clip ->
  score > 0.05 -> the HuggingFace NMS once per class -> best 100): boxes, scores, classes and order
  (tests/test_oracle_hf_nms.py::test_fast_rcnn_inference_equals_a_composition_around_the_independent_nms).
* STILL UNPINNED: what no Detectron2 / torchvision test holds a literal vector for and those libraries (absent here) would
  have to be run for -- the glue of ``fast_rcnn_inference`` against a RUN of Detectron2's function (softmax -> per-class
  decode -> clip -> score > 0.05 -> class-wise NMS -> top-k: every stage is pinned on its own and the whole equals the
  composition above, whose stage order is this repo's reading of the published function), ``Boxes.clip`` /
  ``nonempty`` corner cases, COCOeval's matching rule (simple-sfod_amd/evaluation.py: its AP interpolation equals an independent
  computation through scikit-learn's precision_recall_curve, tests/test_evaluation.py), ColorJitter's parameter sampling; the ResNet-50/101-C4 trunk has no Detectron2 vector but equals
  an independent port -- HuggingFace ``transformers``' ResNet with the stride in the first 1x1 -- on the same weights in
  eval and train mode (tests/test_oracle_r101.py).  Those are restated from their published algorithms and
  anchored on the reference's call sites plus hand-computed known-answer cases (tests/test_oracle_*.py); ``transformers``
  (installed) carries an independent port of torchvision's ``box_iou``: ``oracle.box_ops.pairwise_iou`` / ``box_area`` equal
  it bit for bit (tests/test_oracle_glue.py).  The generator for a full pin is committed (``python -m oracle.gen_golden
  --upstream`` on a machine with detectron2 + torchvision -> tests/golden/upstream_ref.npz, checked by
  tests/test_oracle_upstream.py, which skips until that file exists).
"""
