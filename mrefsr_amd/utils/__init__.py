from .registry import ARCH_REGISTRY, MODEL_REGISTRY, Registry  # noqa: F401
