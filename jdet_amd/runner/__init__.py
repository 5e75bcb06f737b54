from .runner import Runner, synthetic_batch  # noqa: F401
