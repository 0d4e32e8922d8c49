from .repetition import make_repetition_penalty

repetition_penalty_logits_processor = make_repetition_penalty

__all__ = ["make_repetition_penalty", "repetition_penalty_logits_processor"]
