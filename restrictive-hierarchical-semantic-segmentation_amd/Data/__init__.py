from .dataset import TargetEncoder  # noqa: F401
