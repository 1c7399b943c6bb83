"""Throw-away stub (TEST SCAFFOLDING, build container only).

Provides just enough of the ``detectron2`` import surface for
``/root/reference/daod/modeling/meta_arch/vgg.py`` to be loaded BY FILE PATH so that
``oracle/gen_golden.py`` can run the reference's own VGG backbone and record golden
vectors.  It contains no Detectron2 code and is never imported by the product package.
"""
