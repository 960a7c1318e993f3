from .visual_encoders import DrQV2Encoder, RandomShiftAugmentation, SpatialAttention

__all__ = ["DrQV2Encoder", "RandomShiftAugmentation", "SpatialAttention"]
