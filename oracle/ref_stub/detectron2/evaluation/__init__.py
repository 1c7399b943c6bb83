class DatasetEvaluator:
    """Name only (bpc_loss.py imports it, never uses it)."""
