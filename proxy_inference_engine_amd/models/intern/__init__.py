from .ensemble import Model, ModelArgs

__all__ = ["Model", "ModelArgs"]
