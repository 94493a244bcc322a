"""Container-only stand-in for ``tensordict`` so the reference's hot-path modules import.

TEST INFRASTRUCTURE: used only by ``oracle/gen_golden.py`` in the build container to run the
reference's own Python (``/root/reference``) and record golden vectors.  Never imported by the
product package.  Implements just the TensorClass surface the reference touches: attribute
fields, ``clone/to/__getitem__``, ``torch.cat/stack``, pytree flattening (for ``torch.vmap``)
and ``UnbatchedTensor``.
"""
import dataclasses

import torch
from torch.utils import _pytree as pytree


class _Store:
    """Mimics the TensorDict held by a tensorclass (``obj._tensordict._tensordict`` is a dict)."""

    def __init__(self, d):
        self._tensordict = d

    def __getitem__(self, k):
        return self._tensordict[k]

    def __setitem__(self, k, v):
        self._tensordict[k] = v


def _is_leaf_value(v):
    return isinstance(v, (torch.Tensor, _TensorClassBase, UnbatchedTensor))


class _TensorClassMeta(type):
    def __new__(mcs, name, bases, ns, **kw):
        ns = dict(ns)
        defaults = {}
        ann = ns.get("__annotations__", {})
        for k in list(ann):
            if k in ns:
                v = ns.pop(k)
                if isinstance(v, dataclasses.Field):
                    if v.default is not dataclasses.MISSING:
                        defaults[k] = v.default
                else:
                    defaults[k] = v
        cls = super().__new__(mcs, name, bases, ns)
        allann, alldef = {}, {}
        for b in reversed(cls.__mro__):
            allann.update({k: v for k, v in getattr(b, "__annotations__", {}).items() if not k.startswith("_")})
            alldef.update(b.__dict__.get("_own_defaults", {}))
        cls._own_defaults = defaults
        alldef.update(defaults)
        cls._field_names = tuple(allann)
        cls._field_defaults = alldef
        if allann:
            dc = dataclasses.make_dataclass(
                name + "_fields",
                [(k, t, dataclasses.field(default=alldef.get(k, None))) for k, t in allann.items()],
            )
            cls.__dataclass_fields__ = dc.__dataclass_fields__
            pytree.register_pytree_node(cls, cls._tc_flatten, cls._tc_unflatten)
        return cls


class _TensorClassBase(metaclass=_TensorClassMeta):
    def __init__(self, *args, batch_size=None, **kwargs):
        d = {}
        names = type(self)._field_names
        for k, v in zip(names, args):
            d[k] = v
        for k, v in kwargs.items():
            d[k] = v
        for k in names:
            if k not in d:
                d[k] = type(self)._field_defaults.get(k, None)
        object.__setattr__(self, "_tensordict", _Store(d))
        object.__setattr__(self, "_bs", list(batch_size) if batch_size is not None else [])

    # ---- attribute plumbing
    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "_tensordict")._tensordict[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        if name in type(self)._field_names:
            self._tensordict._tensordict[name] = value
        else:
            object.__setattr__(self, name, value)

    @property
    def batch_size(self):
        return torch.Size(self._bs)

    def auto_batch_size_(self):
        for v in self._tensordict._tensordict.values():
            if isinstance(v, torch.Tensor) and v.ndim > 0:
                self._bs = [v.shape[0]]
                break
        return self

    def _map(self, fn):
        out = {}
        for k, v in self._tensordict._tensordict.items():
            if isinstance(v, (torch.Tensor, _TensorClassBase, UnbatchedTensor)):
                out[k] = fn(v)
            elif isinstance(v, tuple) and v and all(isinstance(x, torch.Tensor) or x is None for x in v):
                out[k] = tuple(fn(x) if x is not None else None for x in v)
            else:
                out[k] = v
        new = type(self).__new__(type(self))
        object.__setattr__(new, "_tensordict", _Store(out))
        object.__setattr__(new, "_bs", list(self._bs))
        for extra in ("_device_precomp",):
            if extra in self.__dict__:
                object.__setattr__(new, extra, self.__dict__[extra])
        return new

    def clone(self, recurse=True):
        if recurse:
            return self._map(lambda v: v.clone())
        return self._map(lambda v: v)

    def to(self, *args, **kwargs):
        def mv(v):
            if isinstance(v, torch.Tensor) and not v.is_floating_point():
                # dtype casts only apply to floating tensors
                kw = {k: x for k, x in kwargs.items() if k != "dtype"}
                a = [x for x in args if not isinstance(x, torch.dtype)]
                return v.to(*a, **kw) if (a or kw) else v
            return v.to(*args, **kwargs)

        return self._map(mv)

    def expand(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        new = self._map(lambda v: v.expand(*shape, *v.shape) if isinstance(v, torch.Tensor) else (v if isinstance(v, UnbatchedTensor) else v.expand(*shape)))
        object.__setattr__(new, "_bs", list(shape) + list(self._bs))
        return new

    def contiguous(self):
        return self._map(lambda v: v.contiguous())

    def __getitem__(self, idx):
        new = self._map(lambda v: v if isinstance(v, UnbatchedTensor) else v[idx])
        new.auto_batch_size_() if self._bs else None
        return new

    def __len__(self):
        return self._bs[0]

    # ---- pytree (children = tensor-like fields, everything else rides in the context)
    @classmethod
    def _tc_flatten(cls, obj):
        d = obj._tensordict._tensordict
        keys = obj.__dict__.get("_forced_keys")
        if keys is None:
            keys = [k for k, v in d.items() if _is_leaf_value(v)]
        rest = {k: v for k, v in d.items() if k not in keys}
        return [d[k] for k in keys], (keys, rest, list(obj._bs))

    @classmethod
    def _tc_unflatten(cls, children, ctx):
        keys, rest, bs = ctx
        d = dict(rest)
        d.update(zip(keys, children))
        new = cls.__new__(cls)
        object.__setattr__(new, "_tensordict", _Store(d))
        object.__setattr__(new, "_bs", list(bs))
        object.__setattr__(new, "_forced_keys", list(keys))
        return new

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.cat, torch.stack):
            items = list(args[0])
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            first = items[0]
            out = {}
            for k, v in first._tensordict._tensordict.items():
                if isinstance(v, UnbatchedTensor):
                    out[k] = v
                elif _is_leaf_value(v):
                    out[k] = func([it._tensordict._tensordict[k] for it in items], dim=dim)
                else:
                    out[k] = v
            new = type(first).__new__(type(first))
            object.__setattr__(new, "_tensordict", _Store(out))
            object.__setattr__(new, "_bs", list(first._bs))
            new.auto_batch_size_()
            return new
        return NotImplemented


class _TensorClassFactory(_TensorClassBase):
    """``TensorClass["nocast", "shadow"]`` returns the base class itself."""

    def __class_getitem__(cls, item):
        return _TensorClassBase


TensorClass = _TensorClassFactory


class UnbatchedTensor:
    """Model-constant tensor that ignores batch ops (index / expand / stack keep it as is)."""

    def __init__(self, data=None, batch_size=None):
        self.data = data

    def clone(self):
        return UnbatchedTensor(self.data.clone())

    def to(self, *a, **k):
        return UnbatchedTensor(self.data.to(*a, **k))

    def contiguous(self):
        return self

    def __len__(self):
        return len(self.data)


pytree.register_pytree_node(UnbatchedTensor, lambda u: ([u.data], None), lambda c, _: UnbatchedTensor(c[0]))
