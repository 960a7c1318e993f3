from .visual_encoders import ConvDecoder, DecoderBlock, DrQV2Encoder, RandomShiftAugmentation, SpatialAttention

__all__ = ["ConvDecoder", "DecoderBlock", "DrQV2Encoder", "RandomShiftAugmentation", "SpatialAttention"]
