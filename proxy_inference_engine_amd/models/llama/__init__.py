from .language import Model, ModelArgs

__all__ = ["Model", "ModelArgs"]
