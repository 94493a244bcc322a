"""Minimal tensorclass container with the semantics the reference relies on.

The reference stores Model/Data/Contact in ``tensordict.TensorClass`` subclasses
(``_src/dataclasses.py:85-142``).  tensordict is not a dependency of this package, so the
slice of behaviour callers of ``step`` observe is restated here: attribute fields declared by
annotation, ``replace`` (shallow copy + field swap), ``update_`` (in-place dict update),
``tree_replace`` (dot paths), ``clone(recurse=False)``, ``to``, indexing / ``expand`` over the
leading batch dims, ``torch.stack`` / ``torch.cat``, and ``UnbatchedTensor`` leaves that ignore
batch operations (``ncon``/``nefc``, reference ``types.py:1172-1178``).
"""

from __future__ import annotations

import dataclasses
from typing import Any

import torch


class UnbatchedTensor:
    """A model-constant tensor riding inside a batched container (index/expand/stack keep it)."""

    __slots__ = ("data",)

    def __init__(self, data=None):
        self.data = data

    def clone(self):
        return UnbatchedTensor(self.data.clone())

    def to(self, *args, **kwargs):
        if self.data.is_floating_point():
            return UnbatchedTensor(self.data.to(*args, **kwargs))
        args = [a for a in args if not isinstance(a, torch.dtype)]
        kwargs = {k: v for k, v in kwargs.items() if k != "dtype"}
        return UnbatchedTensor(self.data.to(*args, **kwargs) if (args or kwargs) else self.data)

    def __int__(self):
        return int(self.data)

    def __len__(self):
        return len(self.data)

    def __eq__(self, other):  # value equality: pytree contexts of a step's input and output compare equal (tree_map over both works)
        if not isinstance(other, UnbatchedTensor):
            return NotImplemented
        a, b = self.data, other.data
        if a is b:
            return True
        if a is None or b is None or a.shape != b.shape or a.dtype != b.dtype:
            return False
        return bool(torch.equal(a.detach().cpu(), b.detach().cpu()))

    __hash__ = None

    def __repr__(self):
        return f"UnbatchedTensor({self.data!r})"


# host-side attributes that belong to ONE container instance and are never inherited by containers derived from it
_PRIVATE = ("_fields", "_bs", "_ver", "_child_keys", "_lazy", "_ptab", "_native_cache", "_order", "_op_key", "_op_key_t", "_struct_uid", "_stamp_ts")


def _carve(spec):
    """A leaf of a step's output slab: (slab uint8 1-D, byte offset, byte length, dtype, shape) -> tensor view."""
    slab, off, nbytes, dtype, shape = spec
    return slab.narrow(0, off, nbytes).view(dtype).view(shape)


def _is_node(v) -> bool:
    return isinstance(v, (torch.Tensor, MjTensorClass, UnbatchedTensor))


class _Meta(type):
    def __new__(mcs, name, bases, ns):
        ns = dict(ns)
        own_defaults = {}
        for k in list(ns.get("__annotations__", {})):
            if k in ns:
                own_defaults[k] = ns.pop(k)
        cls = super().__new__(mcs, name, bases, ns)
        ann, defaults = {}, {}
        for b in reversed(cls.__mro__):
            ann.update({k: v for k, v in b.__dict__.get("__annotations__", {}).items() if not k.startswith("_")})
            defaults.update(b.__dict__.get("_own_defaults", {}))
        cls._own_defaults = own_defaults
        defaults.update(own_defaults)
        cls._field_names = tuple(ann)
        cls._field_defaults = defaults
        cls._field_types = ann
        if bases:  # every concrete container is a pytree node, so torch.vmap / tree_map see its tensor leaves
            _register_pytree(cls)
        return cls


def _register_pytree(cls):
    """Children = tensor leaves and nested containers; everything else (UnbatchedTensor, None, python values, the host-side
    attributes) rides in the context.  ``batch_size`` is re-derived from a probe leaf on unflatten, so the per-sample view
    ``torch.vmap`` hands to the mapped function has batch_size () and the stacked result gets the mapped dimension back."""
    from torch.utils import _pytree

    def flatten(obj):
        if torch.compiler.is_compiling():  # Dynamo traces this (torch.vmap flattens its arguments inside the compiled function): no __dict__ access
            keys, children, rest = [], [], {}
            f = obj._fields_traced()
            pinned = obj._child_keys
            for k in f:
                v = f[k]
                if (k in pinned) if pinned is not None else isinstance(v, (torch.Tensor, MjTensorClass)):
                    keys.append(k)
                    children.append(v)
                else:
                    rest[k] = v
            probe = None
            for i in range(len(children)):
                if probe is None and isinstance(children[i], torch.Tensor):
                    probe = i
            event = children[probe].dim() - len(obj._bs) if probe is not None else 0
            return children, (tuple(keys), rest, probe, event, tuple(f), {}, tuple(obj._bs))
        keys, children, rest = [], [], {}
        pinned = obj.__dict__.get("_child_keys")  # set by unflatten: placeholder leaves (pytree's own spec arithmetic) keep the structure
        for k, v in obj._all().items():
            if (k in pinned) if pinned is not None else isinstance(v, (torch.Tensor, MjTensorClass)):
                keys.append(k)
                children.append(v)
            else:
                rest[k] = v
        probe = next((i for i, c in enumerate(children) if isinstance(c, torch.Tensor)), None)
        event = children[probe].dim() - len(obj._bs) if probe is not None else 0
        extra = {k: v for k, v in obj.__dict__.items() if k not in _PRIVATE}
        return children, (tuple(keys), rest, probe, event, tuple(obj._fields), extra, tuple(obj._bs))

    def unflatten(children, ctx):
        keys, rest, probe, event, order, extra, bs = ctx
        vals = dict(rest)
        vals.update(zip(keys, children))
        if torch.compiler.is_compiling():
            if probe is not None and isinstance(children[probe], torch.Tensor):
                t = children[probe]
                bs = tuple(t.shape[: t.dim() - event]) if t.dim() >= event else ()
            new = cls(batch_size=bs, **{k: vals[k] for k in order})
            if not all(isinstance(c, (torch.Tensor, MjTensorClass)) for c in children):
                new._child_keys = frozenset(keys)  # placeholder leaves (pytree's own spec arithmetic) keep the structure
            return new
        new = cls.__new__(cls)
        object.__setattr__(new, "_fields", {k: vals[k] for k in order})
        if probe is not None and isinstance(children[probe], torch.Tensor):
            t = children[probe]
            bs = tuple(t.shape[: t.dim() - event]) if t.dim() >= event else ()
        object.__setattr__(new, "_bs", tuple(bs))
        for k, v in extra.items():
            object.__setattr__(new, k, v)
        if not all(isinstance(c, (torch.Tensor, MjTensorClass)) for c in children):
            object.__setattr__(new, "_child_keys", frozenset(keys))
        new._post_init()
        return new

    _pytree.register_pytree_node(cls, flatten, unflatten)


class MjTensorClass(metaclass=_Meta):
    """Attribute container over a plain dict of leaves plus a leading ``batch_size``."""

    # defaults of the per-instance host attributes (_PRIVATE): code that is traced by Dynamo reads them as plain attributes -- it cannot go
    # through ``__dict__`` -- and finds these when an instance has none of its own
    _lazy = None
    _order = None
    _child_keys = None

    def _post_init(self):
        """Hook run on every freshly built instance (constructor, derived containers, pytree unflatten)."""

    def __init__(self, *args, batch_size=None, **kwargs):
        names = type(self)._field_names
        d = dict(zip(names, args))
        unknown = set(kwargs) - set(names)
        if unknown:
            raise TypeError(f"{type(self).__name__}: unknown fields {sorted(unknown)}")
        d.update(kwargs)
        for k in names:
            if k not in d:
                d[k] = type(self)._field_defaults.get(k)
        object.__setattr__(self, "_fields", d)
        object.__setattr__(self, "_bs", tuple(batch_size) if batch_size is not None else ())
        self._post_init()

    # ---- attribute access ---------------------------------------------------------------
    def __getattr__(self, name):
        if name == "_fields":  # (only reached on a half-built instance: no recursion through the lookups below)
            raise AttributeError(name)
        f = self._fields
        if name in f:
            return f[name]
        lz = self._lazy
        if lz and name in lz:  # a leaf of a step's output slab, carved on first access (forward.py)
            if torch.compiler.is_compiling():
                return _carve(lz[name])  # traced: a view of the slab in the graph, the container is left as it is
            t = _carve(lz.pop(name))
            f[name] = t
            return t
        raise AttributeError(f"{type(self).__name__} has no field {name!r}")

    def __setattr__(self, name, value):
        if name in type(self)._field_names:
            lz = self.__dict__.get("_lazy")
            if lz:
                lz.pop(name, None)
            self._fields[name] = value
            self._touch((name,))
        else:
            object.__setattr__(self, name, value)

    def _all(self):
        """The field dict with every lazily carved leaf materialised, in declaration order."""
        lz = self.__dict__.get("_lazy")
        if lz is not None:  # (possibly emptied by attribute reads, which append to the dict in access order)
            f = self._fields
            for k in list(lz):
                f[k] = _carve(lz.pop(k))
            del self.__dict__["_lazy"]
            order = self.__dict__.get("_order") or type(self)._field_names  # the field order of the container the step was given
            object.__setattr__(self, "_fields", {k: f[k] for k in order if k in f} | {k: v for k, v in f.items() if k not in order})
        return self._fields

    # ---- pickle / copy ----------------------------------------------------------------------
    # A container's process-local attributes (_PRIVATE: the device-pointer table, cached device blobs, the operator key, stamp caches)
    # must not travel with it -- neither into another process (torch.save, a spawn / ParallelEnv worker) nor into a copy made by
    # copy.copy / copy.deepcopy, which would otherwise duplicate ``__dict__`` as it is (a copy with the original's ``_op_key`` would be
    # stepped with the ORIGINAL's values under torch.vmap / torch.compile).  pickle, copy and deepcopy all go through this pair;
    # ``_post_init`` runs on the result as it does on every other freshly built instance.
    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in _PRIVATE}
        state["_fields"] = dict(self._all())  # lazily carved leaves of a step's output are materialised first
        state["_bs"] = tuple(self._bs)
        pinned = self.__dict__.get("_child_keys")
        if pinned is not None:
            state["_child_keys"] = pinned
        return state

    def __setstate__(self, state):
        for k, v in state.items():
            object.__setattr__(self, k, v)
        self._post_init()

    def _touch(self, names=None):
        """Bumps the leaf-set version and tells the container's device-pointer table (forward.py) which leaves changed."""
        object.__setattr__(self, "_ver", self.__dict__.get("_ver", 0) + 1)
        tab = self.__dict__.get("_ptab")
        if tab is not None:
            tab.mark(names)

    @classmethod
    def fields(cls):
        return tuple(dataclasses.field() for _ in cls._field_names)

    @classmethod
    def field_names(cls):
        return cls._field_names

    @property
    def batch_size(self):
        return torch.Size(self._bs)

    def items(self):
        return self._all().items()

    # ---- functional updates (reference dataclasses.py:101-127) ---------------------------
    def _new(self, d, bs=None):
        new = type(self).__new__(type(self))
        object.__setattr__(new, "_fields", d)
        object.__setattr__(new, "_bs", tuple(self._bs if bs is None else bs))
        for k, v in self.__dict__.items():
            if k not in _PRIVATE:
                object.__setattr__(new, k, v)
        new._post_init()
        return new

    def _fields_traced(self):
        """Every field with the lazily carved leaves as views, in declaration order, WITHOUT touching the container (Dynamo-traceable)."""
        f, lz = self._fields, self._lazy
        if not lz:
            return f
        names = self._order or type(self)._field_names
        out = {}
        for k in names:
            if k in f:
                out[k] = f[k]
            elif k in lz:
                out[k] = _carve(lz[k])
        for k in f:
            if k not in out:
                out[k] = f[k]
        return out

    def _replace_traced(self, kwargs):
        """``replace`` for code traced by Dynamo: plain dict / constructor operations only (Data and Contact carry no extra host attributes)."""
        f = dict(self._fields_traced())
        f.update(kwargs)
        return type(self)(batch_size=self._bs, **f)

    def map_tensors(self, fn):
        """Applies ``fn`` to every tensor leaf, recursing into nested containers; other values are kept."""
        out = {}
        for k, v in self._all().items():
            if isinstance(v, MjTensorClass):
                out[k] = v.map_tensors(fn)
            elif isinstance(v, torch.Tensor):
                out[k] = fn(v)
            else:
                out[k] = v
        return self._new(out)

    def _map(self, fn, bs=None):
        return self._new({k: (fn(v) if _is_node(v) else v) for k, v in self._all().items()}, bs)

    def clone(self, recurse: bool = True):
        if recurse:
            return self._map(lambda v: v.clone())
        new = self._new(dict(self._fields))
        lz = self.__dict__.get("_lazy")
        if lz:
            object.__setattr__(new, "_lazy", dict(lz))  # same slab regions, carved independently
            if "_order" in self.__dict__:  # the field order the lazy leaves re-enter in belongs with them (and with nothing else: _PRIVATE)
                object.__setattr__(new, "_order", self.__dict__["_order"])
        return new

    def replace(self, **kwargs: Any):
        if torch.compiler.is_compiling():
            return self._replace_traced(kwargs)
        new = self.clone(recurse=False)
        lz = new.__dict__.get("_lazy")
        if lz:
            for k in kwargs:
                lz.pop(k, None)
        new._fields.update(kwargs)
        tab = self.__dict__.get("_ptab")
        if tab is not None:  # the pointer table follows the container: only the replaced leaves are looked at again
            object.__setattr__(new, "_ptab", tab.child(kwargs))
        return new

    def update_(self, **kwargs: Any):
        lz = self.__dict__.get("_lazy")
        if lz:
            for k in kwargs:
                lz.pop(k, None)
        self._fields.update(kwargs)
        self._touch(kwargs)
        return self

    def tree_replace(self, params: dict):
        new = self
        for path, val in params.items():
            keys = path.split(".")
            new = new._tree_replace(keys, val)
        return new

    def _tree_replace(self, keys, val):
        if len(keys) == 1:
            return self.replace(**{keys[0]: val})
        return self.replace(**{keys[0]: getattr(self, keys[0])._tree_replace(keys[1:], val)})

    # ---- device / dtype --------------------------------------------------------------------
    def to(self, *args, **kwargs):
        def mv(v):
            if isinstance(v, torch.Tensor) and not v.is_floating_point():
                a = [x for x in args if not isinstance(x, torch.dtype)]
                kw = {k: x for k, x in kwargs.items() if k != "dtype"}
                return v.to(*a, **kw) if (a or kw) else v
            return v.to(*args, **kwargs)

        return self._map(mv)

    def cpu(self):
        return self.to("cpu")

    def contiguous(self):
        return self._map(lambda v: v if isinstance(v, UnbatchedTensor) else v.contiguous())

    # ---- batch-dim operations ----------------------------------------------------------------
    def expand(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        nb = len(self._bs)
        lead = tuple(shape[: len(shape) - nb]) if nb else tuple(shape)

        def ex(v):
            if isinstance(v, UnbatchedTensor):
                return v
            if isinstance(v, MjTensorClass):
                return v.expand(*lead, *v._bs)
            return v.expand(*lead, *v.shape)

        return self._map(ex, bs=lead + tuple(self._bs))

    def __getitem__(self, idx):
        def ix(v):
            if isinstance(v, UnbatchedTensor):
                return v
            return v[idx]

        probe = torch.empty(self._bs)[idx] if self._bs else None
        bs = tuple(probe.shape) if probe is not None else ()
        return self._map(ix, bs=bs)

    def __setitem__(self, idx, value):
        for k, v in self._all().items():
            if isinstance(v, UnbatchedTensor) or not _is_node(v):
                continue
            src = getattr(value, k)
            if isinstance(v, MjTensorClass):
                v[idx] = src
            else:
                v[idx] = src

    def __len__(self):
        return self._bs[0]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.stack, torch.cat):
            items = list(args[0])
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            first = items[0]
            out = {}
            for it in items:
                it._all()
            for k, v in first._fields.items():
                if isinstance(v, UnbatchedTensor) or not _is_node(v):
                    out[k] = v
                else:
                    out[k] = func([it._fields[k] for it in items], dim=dim)
            bs = list(first._bs)
            if func is torch.stack:
                bs.insert(dim, len(items))
            else:
                bs[dim] = sum(it._bs[dim] for it in items)
            return first._new(out, bs)
        return NotImplemented

    def __repr__(self):
        return f"{type(self).__name__}(batch_size={list(self._bs)}, fields={len(self._fields) + len(self.__dict__.get('_lazy') or ())})"
