from .utils import load, load_model, synthetic_checkpoint

__all__ = ["load", "load_model", "synthetic_checkpoint"]
