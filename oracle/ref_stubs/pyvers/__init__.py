"""Container-only stand-in for ``pyvers.implement_for`` (version dispatch on the stub mujoco 3.5.0)."""
_VERSION = (3, 5, 0)


def _parse(v):
    return tuple(int(x) for x in v.split("."))


def _matches(from_version, to_version):
    if from_version is not None and _VERSION < _parse(from_version):
        return False
    if to_version is not None and _VERSION >= _parse(to_version):
        return False
    return True


class _Dispatch:
    def __init__(self, fn=None):
        self.impl = fn
        self.fallback = fn

    def register(self, from_version=None, to_version=None):
        def deco(fn):
            if _matches(from_version, to_version):
                self.impl = fn
            return self

        return deco

    def __call__(self, *a, **k):
        return self.impl(*a, **k)


_REGISTRY = {}


def implement_for(module, from_version=None, to_version=None):
    def deco(fn):
        if from_version is None and to_version is None:
            return _Dispatch(fn)
        # same-name redefinition style (sensor.py): keep the first matching implementation
        key = (fn.__module__, fn.__qualname__)
        d = _REGISTRY.setdefault(key, _Dispatch(None))
        if _matches(from_version, to_version):
            d.impl = fn
        return d

    return deco
