from .extractors import OrientedSingleRoIExtractor, RboxSingleRoIExtractor, SingleRoIExtractor  # noqa: F401
