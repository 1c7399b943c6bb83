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
* PINNED against the reference itself (imported in the build container, outputs
  committed under ``tests/golden/`` by ``oracle/gen_golden.py``): the VGG16-BN backbone
  (``vgg.py``, loaded by file path behind ``oracle/ref_stub``) forward, train-mode BN
  running stats and input gradient; ``dann.py`` discriminators + gradient reversal.
* PARITY UNPINNED for everything that lives in Detectron2 / torchvision (anchors,
  Box2BoxTransform, Matcher, NMS, ROIAlign, RPN / Fast R-CNN losses and inference):
  the reference ships no tests or golden vectors and those libraries are absent, so
  those functions are restated from their published algorithms and anchored on the
  reference's call sites plus hand-computed known-answer cases (tests/test_oracle_*.py).
"""
