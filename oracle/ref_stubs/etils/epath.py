from pathlib import Path  # noqa: F401


def resource_path(_):
    raise RuntimeError("etils stub")
