def detector_postprocess(*a, **k):
    raise NotImplementedError("imported by bpc_loss.py, not called")
