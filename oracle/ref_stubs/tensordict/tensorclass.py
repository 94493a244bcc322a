from tensordict import _TensorClassMeta  # noqa: F401
