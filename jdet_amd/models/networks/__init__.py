from .s2anet import S2ANet  # noqa: F401
