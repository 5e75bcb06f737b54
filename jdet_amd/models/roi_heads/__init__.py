from .s2anet_head import AlignConv, S2ANetHead, bbox_decode  # noqa: F401
