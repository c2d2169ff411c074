"""NumPy stand-in for the part of TensorFlow that GPflow touches (see ../README.md).  Test infrastructure only.

Eager semantics only: a `Tensor` wraps a `numpy.ndarray`; every op unwraps, calls NumPy / SciPy, wraps.  No graphs, no
gradients.  Linear algebra maps to LAPACK through NumPy / SciPy (cholesky = dpotrf, triangular_solve = dtrtrs)."""
from __future__ import annotations

import builtins
import sys
import types

import numpy as np
import scipy.linalg as _sla
import scipy.special as _ssp

__version__ = "2.11.1"  # the reference's "prod" CI pin (.circleci/config.yml); GPflow only compares it with Version(...)
newaxis = None


# ---- dtypes -------------------------------------------------------------------------------------------------------------
class DType:
    def __init__(self, np_dtype):
        self._np = np.dtype(np_dtype)

    @property
    def as_numpy_dtype(self):
        return self._np.type

    @property
    def name(self):
        return self._np.name

    @property
    def is_integer(self):
        return np.issubdtype(self._np, np.integer)

    @property
    def is_floating(self):
        return np.issubdtype(self._np, np.floating)

    @property
    def is_bool(self):
        return self._np == np.bool_

    @property
    def base_dtype(self):
        return self

    @property
    def min(self):
        return np.finfo(self._np).min if self.is_floating else np.iinfo(self._np).min

    @property
    def max(self):
        return np.finfo(self._np).max if self.is_floating else np.iinfo(self._np).max

    def __eq__(self, other):
        try:
            return self._np == _npdtype(other)
        except TypeError:
            return False

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._np)

    def __repr__(self):
        return f"tf.{self._np.name}"


def _npdtype(d):
    if d is None:
        return None
    if isinstance(d, DType):
        return d._np
    return np.dtype(d)


float16, float32, float64 = DType(np.float16), DType(np.float32), DType(np.float64)
int8, int16, int32, int64 = DType(np.int8), DType(np.int16), DType(np.int32), DType(np.int64)
uint8 = DType(np.uint8)
bool = DType(np.bool_)  # noqa: A001  (tf.bool)
string = DType(np.str_)
double = float64


def as_dtype(d):
    return d if isinstance(d, DType) else DType(d)


class TensorShape(tuple):
    def __new__(cls, dims=()):
        if dims is None:
            dims = ()
        return super().__new__(cls, tuple(dims))

    @property
    def ndims(self):
        return len(self)

    @property
    def rank(self):
        return len(self)

    def as_list(self):
        return list(self)

    def is_fully_defined(self):
        return all(d is not None for d in self)

    def num_elements(self):
        return int(np.prod(self)) if len(self) else 1

    def __getitem__(self, k):
        r = tuple.__getitem__(self, k)
        return TensorShape(r) if isinstance(k, slice) else r

    def __add__(self, other):
        return TensorShape(tuple(self) + tuple(other))

    def is_compatible_with(self, other):
        other = tuple(other)
        return len(other) == len(self) and all(a is None or b is None or a == b for a, b in zip(self, other))


# ---- tensors ------------------------------------------------------------------------------------------------------------
def _np(x, dtype=None):
    """unwrap anything tensor-like to a numpy array"""
    if isinstance(x, Tensor):
        a = x._value()
    elif hasattr(x, "__tf_tensor__"):
        a = x.__tf_tensor__()
    elif isinstance(x, (list, tuple)):
        a = np.array([_np(e) for e in x]) if any(isinstance(e, (Tensor, list, tuple, np.ndarray)) or hasattr(e, "__tf_tensor__") for e in x) else np.array(x)
    else:
        a = np.asarray(x)
    if dtype is not None:
        a = a.astype(_npdtype(dtype), copy=False)
    return a


def _wrap(a):
    if isinstance(a, Tensor):
        return a
    return Tensor(np.asarray(a))


class Tensor(np.lib.mixins.NDArrayOperatorsMixin):
    __array_priority__ = 100

    def __init__(self, value):
        self._v = np.asarray(value)

    def _value(self):
        return self._v

    # --- numpy protocol
    def __array__(self, dtype=None, copy=None):
        v = self._value()
        return v.astype(dtype) if dtype is not None else v

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        ins = [_np(i) if isinstance(i, Tensor) or hasattr(i, "__tf_tensor__") else i for i in inputs]
        if "out" in kwargs:
            return NotImplemented
        r = getattr(ufunc, method)(*ins, **kwargs)
        if isinstance(r, tuple):
            return tuple(_wrap(x) for x in r)
        return _wrap(r)

    # `x += y` on a tf.Tensor rebinds the name (tensors are immutable)
    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o

    def __matmul__(self, other):
        return _wrap(np.matmul(self._value(), _np(other)))

    def __rmatmul__(self, other):
        return _wrap(np.matmul(_np(other), self._value()))

    # --- tf.Tensor surface
    @property
    def dtype(self):
        return DType(self._value().dtype)

    @property
    def shape(self):
        return TensorShape(self._value().shape)

    @property
    def ndim(self):
        return self._value().ndim

    def get_shape(self):
        return self.shape

    def numpy(self):
        v = self._value()
        return v.copy() if v.ndim else v[()]

    def __getitem__(self, k):
        if isinstance(k, tuple):
            k = tuple(_np(i) if isinstance(i, Tensor) else i for i in k)
        elif isinstance(k, Tensor):
            k = _np(k)
        return _wrap(self._value()[k])

    def __len__(self):
        return len(self._value())

    def __iter__(self):
        for i in builtins.range(len(self._value())):
            yield self[i]

    def __float__(self):
        return float(self._value())

    def __int__(self):
        return int(self._value())

    def __index__(self):
        return int(self._value())

    def __bool__(self):
        return builtins.bool(self._value())

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return f"<shim tf.Tensor shape={tuple(self.shape)} dtype={self.dtype.name} value={self._value()!r}>"

    def ref(self):
        return _Ref(self)

    def set_shape(self, shape):
        pass

    @property
    def T(self):
        return _wrap(self._value().T)


class _Ref:
    def __init__(self, t):
        self._t = t

    def deref(self):
        return self._t

    def __hash__(self):
        return id(self._t)

    def __eq__(self, other):
        return isinstance(other, _Ref) and other._t is self._t


class Variable(Tensor):
    def __init__(self, initial_value=None, trainable=True, name=None, dtype=None, shape=None, **kw):
        if callable(initial_value):
            initial_value = initial_value()
        v = _np(initial_value, dtype)
        super().__init__(np.array(v, copy=True))
        self._trainable = True if trainable is None else builtins.bool(trainable)
        self.name = (name or "Variable") + ":0"

    @property
    def trainable(self):
        return self._trainable

    def assign(self, value, use_locking=False, name=None, read_value=True):
        v = _np(value).astype(self._v.dtype)
        self._v = np.array(np.broadcast_to(v, self._v.shape) if v.shape != self._v.shape and v.size == 1 else v, copy=True)
        return self

    def assign_add(self, delta, **kw):
        return self.assign(self._v + _np(delta))

    def assign_sub(self, delta, **kw):
        return self.assign(self._v - _np(delta))

    def value(self):
        return _wrap(self._v)

    def read_value(self):
        return _wrap(self._v)

    @property
    def handle(self):
        return self


def is_tensor(x):
    return isinstance(x, Tensor) or hasattr(x, "__tf_tensor__")


def convert_to_tensor(value, dtype=None, dtype_hint=None, name=None):
    # (Python floats become float64 here, float32 in TensorFlow: GPflow always passes default_float() where it matters,
    #  and NumPy would silently up-cast a mixed product that TensorFlow refuses)
    a = _np(value)
    if dtype is not None:
        a = a.astype(_npdtype(dtype))
    elif isinstance(value, int) and not isinstance(value, builtins.bool):
        a = a.astype(np.int32)
    return Tensor(a)


def constant(value, dtype=None, shape=None, name=None):
    t = convert_to_tensor(value, dtype=dtype)
    if shape is not None:
        t = Tensor(np.broadcast_to(t._v, tuple(shape)).copy())
    return t


def cast(x, dtype, name=None):
    return Tensor(_np(x).astype(_npdtype(dtype)))


def identity(x, name=None):
    return _wrap(_np(x))


def stop_gradient(x, name=None):
    return _wrap(_np(x))


def ensure_shape(x, shape, name=None):
    return _wrap(_np(x))


# ---- shapes ---------------------------------------------------------------------------------------------------------------
def _shp(s):
    if isinstance(s, (int, np.integer)):
        return (int(s),)
    a = _np(s)
    return tuple(int(v) for v in np.atleast_1d(a).tolist())


def shape(x, out_type=None, name=None):
    return Tensor(np.array(_np(x).shape, dtype=_npdtype(out_type) if out_type is not None else np.int32))


def rank(x, name=None):
    return Tensor(np.array(_np(x).ndim, dtype=np.int32))


def size(x, out_type=None, name=None):
    return Tensor(np.array(_np(x).size, dtype=np.int32))


def reshape(x, shape, name=None):  # noqa: A002
    return Tensor(np.reshape(_np(x), _shp(shape)))


def transpose(a, perm=None, conjugate=False, name=None):
    a = _np(a)
    if perm is None:
        return Tensor(np.transpose(a))
    return Tensor(np.transpose(a, _shp(perm)))


def expand_dims(x, axis, name=None):
    return Tensor(np.expand_dims(_np(x), int(_np(axis))))


def squeeze(x, axis=None, name=None):
    if axis is not None:
        axis = tuple(_shp(axis))
    return Tensor(np.squeeze(_np(x), axis=axis))


def concat(values, axis, name=None):
    return Tensor(np.concatenate([np.atleast_1d(_np(v)) for v in values], axis=int(_np(axis))))


def stack(values, axis=0, name=None):
    return Tensor(np.stack([_np(v) for v in values], axis=int(axis)))


def unstack(value, num=None, axis=0, name=None):
    v = np.moveaxis(_np(value), axis, 0)
    return [Tensor(v[i]) for i in range(v.shape[0])]


def tile(x, multiples, name=None):
    return Tensor(np.tile(_np(x), _shp(multiples)))


def broadcast_to(x, shape, name=None):  # noqa: A002
    return Tensor(np.broadcast_to(_np(x), _shp(shape)).copy())


def zeros(shape, dtype=float32, name=None):  # noqa: A002
    return Tensor(np.zeros(_shp(shape), dtype=_npdtype(dtype)))


def ones(shape, dtype=float32, name=None):  # noqa: A002
    return Tensor(np.ones(_shp(shape), dtype=_npdtype(dtype)))


def fill(dims, value, name=None):
    v = _np(value)
    return Tensor(np.full(_shp(dims), v, dtype=v.dtype))


def zeros_like(x, dtype=None, name=None):
    return Tensor(np.zeros_like(_np(x), dtype=_npdtype(dtype)))


def ones_like(x, dtype=None, name=None):
    return Tensor(np.ones_like(_np(x), dtype=_npdtype(dtype)))


def eye(num_rows, num_columns=None, batch_shape=None, dtype=float32, name=None):
    n = int(_np(num_rows))
    m = n if num_columns is None else int(_np(num_columns))
    e = np.eye(n, m, dtype=_npdtype(dtype))
    if batch_shape is not None:
        e = np.broadcast_to(e, _shp(batch_shape) + e.shape).copy()
    return Tensor(e)


def range(start, limit=None, delta=1, dtype=None, name=None):  # noqa: A001
    if limit is None:
        start, limit = 0, start
    a = np.arange(_np(start), _np(limit), _np(delta))
    if dtype is not None:
        a = a.astype(_npdtype(dtype))
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    return Tensor(a)


def gather(params, indices, validate_indices=None, axis=None, batch_dims=0, name=None):
    return Tensor(np.take(_np(params), _np(indices), axis=0 if axis is None else int(_np(axis))))


def where(condition, x=None, y=None, name=None):
    if x is None:
        return Tensor(np.argwhere(_np(condition)))
    return Tensor(np.where(_np(condition), _np(x), _np(y)))


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    v = _np(value)
    if isinstance(num_or_size_splits, (int, np.integer)):
        parts = np.split(v, int(num_or_size_splits), axis=axis)
    else:
        sizes = _shp(num_or_size_splits)
        parts = np.split(v, np.cumsum(sizes)[:-1], axis=axis)
    return [Tensor(p) for p in parts]


def one_hot(indices, depth, dtype=float32, **kw):
    return Tensor(np.eye(int(depth), dtype=_npdtype(dtype))[_np(indices)])


def meshgrid(*args, indexing="xy", **kw):
    return [Tensor(g) for g in np.meshgrid(*[_np(a) for a in args], indexing=indexing)]


def sort(values, axis=-1, direction="ASCENDING", name=None):
    s = np.sort(_np(values), axis=axis)
    return Tensor(s if direction == "ASCENDING" else np.flip(s, axis=axis))


def argmax(x, axis=None, output_type=int64, name=None):
    return Tensor(np.argmax(_np(x), axis=axis).astype(_npdtype(output_type)))


def unique(x, out_idx=int32, name=None):
    xv = _np(x)
    _, first, inv = np.unique(xv, return_index=True, return_inverse=True)
    order = np.argsort(first)
    remap = np.empty_like(order)
    remap[order] = np.arange(len(order))
    return Tensor(xv[np.sort(first)]), Tensor(remap[inv].astype(_npdtype(out_idx)))


def dynamic_partition(data, partitions, num_partitions, name=None):
    d, p = _np(data), _np(partitions)
    return [Tensor(d[p == i]) for i in builtins.range(num_partitions)]


def dynamic_stitch(indices, data, name=None):
    idx = [_np(i) for i in indices]
    dat = [_np(d) for d in data]
    n = builtins.max(int(i.max()) for i in idx if i.size) + 1
    first = next(d for d in dat if d.size)
    out = np.zeros((n,) + first.shape[1:], dtype=first.dtype)
    for i, d in zip(idx, dat):
        out[i] = d
    return Tensor(out)


# ---- elementwise / reductions ------------------------------------------------------------------------------------------
def _unary(f):
    def op(x, name=None):
        return Tensor(f(_np(x)))
    return op


def _binary(f):
    def op(x, y, name=None):
        return Tensor(f(_np(x), _np(y)))
    return op


square = _unary(np.square)
sqrt = _unary(np.sqrt)
exp = _unary(np.exp)
abs = _unary(np.abs)  # noqa: A001
sin, cos, acos, tanh = _unary(np.sin), _unary(np.cos), _unary(np.arccos), _unary(np.tanh)
sigmoid = _unary(_ssp.expit)
negative = _unary(np.negative)
sign = _unary(np.sign)
add, subtract, multiply, divide = _binary(np.add), _binary(np.subtract), _binary(np.multiply), _binary(np.divide)
maximum, minimum = _binary(np.maximum), _binary(np.minimum)
pow = _binary(np.power)  # noqa: A001
equal, less, greater = _binary(np.equal), _binary(np.less), _binary(np.greater)
not_equal, less_equal, greater_equal = _binary(np.not_equal), _binary(np.less_equal), _binary(np.greater_equal)
logical_and, logical_or = _binary(np.logical_and), _binary(np.logical_or)
logical_not = _unary(np.logical_not)


def softplus(x, name=None):
    return Tensor(np.logaddexp(0.0, _np(x)))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return Tensor(np.clip(_np(t), _np(clip_value_min), _np(clip_value_max)))


def add_n(inputs, name=None):
    out = _np(inputs[0])
    for x in inputs[1:]:
        out = out + _np(x)
    return Tensor(out)


def _axis(axis):
    if axis is None:
        return None
    a = _np(axis)
    return int(a) if a.ndim == 0 else tuple(int(v) for v in a)


def reduce_sum(x, axis=None, keepdims=False, name=None):
    return Tensor(np.sum(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_mean(x, axis=None, keepdims=False, name=None):
    return Tensor(np.mean(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_prod(x, axis=None, keepdims=False, name=None):
    return Tensor(np.prod(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_max(x, axis=None, keepdims=False, name=None):
    return Tensor(np.max(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_min(x, axis=None, keepdims=False, name=None):
    return Tensor(np.min(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_all(x, axis=None, keepdims=False, name=None):
    return Tensor(np.all(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_any(x, axis=None, keepdims=False, name=None):
    return Tensor(np.any(_np(x), axis=_axis(axis), keepdims=keepdims))


def reduce_logsumexp(x, axis=None, keepdims=False, name=None):
    return Tensor(_ssp.logsumexp(_np(x), axis=_axis(axis), keepdims=keepdims))


def matmul(a, b, transpose_a=False, transpose_b=False, adjoint_a=False, adjoint_b=False, name=None, **kw):
    a, b = _np(a), _np(b)
    if transpose_a or adjoint_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b or adjoint_b:
        b = np.swapaxes(b, -1, -2)
    return Tensor(np.matmul(a, b))


def tensordot(a, b, axes, name=None):
    if not isinstance(axes, (int, np.integer)):
        axes = [list(np.atleast_1d(_np(ax))) for ax in axes]
    return Tensor(np.tensordot(_np(a), _np(b), axes=axes))


def einsum(equation, *inputs, **kw):
    return Tensor(np.einsum(equation, *[_np(i) for i in inputs]))


def map_fn(fn, elems, fn_output_signature=None, dtype=None, **kw):
    if isinstance(elems, (tuple, list)):
        arrs = [_np(e) for e in elems]
        n = arrs[0].shape[0]
        outs = [fn(type(elems)(Tensor(a[i]) for a in arrs)) for i in builtins.range(n)]
    else:
        arr = _np(elems)
        outs = [fn(Tensor(arr[i])) for i in builtins.range(arr.shape[0])]
    if isinstance(outs[0], (tuple, list)):
        return type(outs[0])(Tensor(np.stack([_np(o[j]) for o in outs])) for j in builtins.range(len(outs[0])))
    return Tensor(np.stack([_np(o) for o in outs]))


def cond(pred, true_fn=None, false_fn=None, name=None):
    return true_fn() if builtins.bool(_np(pred)) else false_fn()


def while_loop(cond, body, loop_vars, **kw):  # noqa: A002
    vars_ = tuple(loop_vars)
    while builtins.bool(_np(cond(*vars_))):
        vars_ = tuple(body(*vars_))
    return vars_


def function(func=None, **kw):
    if func is None:
        return lambda f: f
    return func


def custom_gradient(f):
    def wrapped(*a, **k):
        return f(*a, **k)[0]
    return wrapped


class _NullCtx:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __call__(self, f):
        return f


name_scope = _NullCtx
init_scope = _NullCtx
control_dependencies = _NullCtx


class GradientTape(_NullCtx):
    def watch(self, *a):
        pass

    def gradient(self, *a, **k):
        raise NotImplementedError("the TensorFlow stand-in has no automatic differentiation")


class UnconnectedGradients:
    NONE = "none"
    ZERO = "zero"


# ---- tf.Module -------------------------------------------------------------------------------------------------------------
class Module:
    def __init__(self, name=None):
        self._name = name or type(self).__name__

    @property
    def name(self):
        return getattr(self, "_name", type(self).__name__)

    @property
    def name_scope(self):
        return _NullCtx()

    @classmethod
    def with_name_scope(cls, method):
        return method

    def _flatten(self, recursive=True, predicate=None, with_path=False, expand_composites=False, _seen=None, _path=()):
        seen = _seen if _seen is not None else set()
        out = []

        def visit(obj, path):
            if id(obj) in seen:
                return
            if predicate is None or predicate(obj):
                seen.add(id(obj))
                out.append((path, obj) if with_path else obj)
                if not isinstance(obj, Module):
                    return
            if isinstance(obj, Module):
                if obj is not self and not recursive:
                    return
                seen.add(id(obj))
                for k in sorted(vars(obj)):
                    if k.startswith("_tf_") or k in ("_name",):
                        continue
                    visit(vars(obj)[k], path + (k,))
            elif isinstance(obj, (list, tuple)):
                for i, e in enumerate(obj):
                    visit(e, path + (i,))
            elif isinstance(obj, dict):
                for k in sorted(obj, key=str):
                    visit(obj[k], path + (k,))
            elif hasattr(obj, "_tf_composite_parts"):
                for k, e in obj._tf_composite_parts():
                    visit(e, path + (k,))

        for k in sorted(vars(self)):
            if k in ("_name",):
                continue
            visit(vars(self)[k], _path + (k,))
        return out

    @property
    def submodules(self):
        return tuple(self._flatten(predicate=lambda o: isinstance(o, Module)))

    @property
    def variables(self):
        return tuple(self._flatten(predicate=lambda o: isinstance(o, Variable), expand_composites=True))

    @property
    def trainable_variables(self):
        return tuple(v for v in self.variables if v.trainable)

    @property
    def non_trainable_variables(self):
        return tuple(v for v in self.variables if not v.trainable)


# ---- permissive placeholders for everything that is only referenced (never executed) on the hot path ---------------------
class _Dummy:
    """Absorbs attribute access, calls and subclassing: enough for `class X(tf.keras.optimizers.Optimizer)` or
    `tf.summary.scalar` to be REFERENCED at import time."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy

    def __mro_entries__(self, bases):
        return (_DummyBase,)


class _DummyBase:
    def __init__(self, *a, **k):
        pass


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _DummyBase if name[:1].isupper() else _Dummy()


def _submodule(name, **attrs):
    m = _DummyModule(__name__ + "." + name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[m.__name__] = m
    return m


# ---- tf.linalg -----------------------------------------------------------------------------------------------------------
class _InvalidArgumentError(Exception):
    pass


def _cholesky(x, name=None):
    a = _np(x)
    try:
        return Tensor(np.linalg.cholesky(a))
    except np.linalg.LinAlgError as e:  # TF (CPU): InvalidArgumentError "Cholesky decomposition was not successful"
        raise _InvalidArgumentError("Cholesky decomposition was not successful. The input might not be valid.") from e


def _triangular_solve(matrix, rhs, lower=True, adjoint=False, name=None):
    m, r = _np(matrix), _np(rhs)
    bshape = np.broadcast_shapes(m.shape[:-2], r.shape[:-2])
    m = np.broadcast_to(m, bshape + m.shape[-2:])
    r = np.broadcast_to(r, bshape + r.shape[-2:])
    out = np.empty(r.shape, dtype=np.result_type(m, r))
    for idx in np.ndindex(*bshape):
        out[idx] = _sla.solve_triangular(m[idx], r[idx], lower=lower, trans="T" if adjoint else "N", check_finite=False)
    return Tensor(out)


def _cholesky_solve(chol, rhs, name=None):
    y = _triangular_solve(chol, rhs, lower=True)
    return _triangular_solve(chol, y, lower=True, adjoint=True)


def _band_part(x, num_lower, num_upper, name=None):
    a = _np(x)
    m, n = a.shape[-2:]
    i, j = np.arange(m)[:, None], np.arange(n)[None, :]
    keep = np.ones((m, n), dtype=builtins.bool)
    if num_lower >= 0:
        keep &= (i - j) <= num_lower
    if num_upper >= 0:
        keep &= (j - i) <= num_upper
    return Tensor(np.where(keep, a, np.zeros((), dtype=a.dtype)))


def _diag(diagonal, name=None, k=0, **kw):
    d = _np(diagonal)
    out = np.zeros(d.shape + d.shape[-1:], dtype=d.dtype)
    idx = np.arange(d.shape[-1])
    out[..., idx, idx] = d
    return Tensor(out)


def _diag_part(x, name=None, k=0, **kw):
    return Tensor(np.diagonal(_np(x), axis1=-2, axis2=-1).copy())


def _set_diag(x, diagonal, name=None, k=0, **kw):
    a = np.array(_np(x), copy=True)
    idx = np.arange(builtins.min(a.shape[-2:]))
    a[..., idx, idx] = _np(diagonal)
    return Tensor(a)


def _adjoint(x, name=None):
    return Tensor(np.swapaxes(_np(x), -1, -2))


def _trace(x, name=None):
    return Tensor(np.trace(_np(x), axis1=-2, axis2=-1))


def _matvec(a, b, transpose_a=False, **kw):
    a = _np(a)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    return Tensor(np.einsum("...ij,...j->...i", a, _np(b)))


def _eigh(x, name=None):
    w, v = np.linalg.eigh(_np(x))
    return Tensor(w), Tensor(v)


linalg = _submodule(
    "linalg", cholesky=_cholesky, triangular_solve=_triangular_solve, cholesky_solve=_cholesky_solve, matmul=matmul,
    band_part=_band_part, diag=_diag, diag_part=_diag_part, set_diag=_set_diag, adjoint=_adjoint,
    matrix_transpose=_adjoint, trace=_trace, matvec=_matvec, eigh=_eigh, eye=eye, einsum=einsum, tensordot=tensordot,
    inv=lambda x, **k: Tensor(np.linalg.inv(_np(x))), det=lambda x, **k: Tensor(np.linalg.det(_np(x))),
    logdet=lambda x, **k: Tensor(np.linalg.slogdet(_np(x))[1]),
    slogdet=lambda x, **k: tuple(Tensor(v) for v in np.linalg.slogdet(_np(x))),
    solve=lambda a, b, **k: Tensor(np.linalg.solve(_np(a), _np(b))),
)

math = _submodule(
    "math", log=_unary(np.log), exp=exp, sqrt=sqrt, square=square, lgamma=_unary(_ssp.gammaln), erf=_unary(_ssp.erf),
    erfc=_unary(_ssp.erfc), log1p=_unary(np.log1p), expm1=_unary(np.expm1), reduce_sum=reduce_sum, reduce_prod=reduce_prod,
    reduce_mean=reduce_mean, reduce_logsumexp=reduce_logsumexp, reduce_max=reduce_max, reduce_min=reduce_min,
    softplus=softplus, sigmoid=sigmoid, abs=abs, maximum=maximum, minimum=minimum, add=add, multiply=multiply,
    subtract=subtract, divide=divide, pow=pow, tanh=tanh, sin=sin, cos=cos, sign=sign, negative=negative,
    is_nan=_unary(np.isnan), is_inf=_unary(np.isinf), is_finite=_unary(np.isfinite), digamma=_unary(_ssp.digamma),
    rsqrt=lambda x, **k: Tensor(1.0 / np.sqrt(_np(x))), reciprocal=lambda x, **k: Tensor(1.0 / _np(x)),
    cumsum=lambda x, axis=0, **k: Tensor(np.cumsum(_np(x), axis=axis)), equal=equal, less=less, greater=greater,
    logical_and=logical_and, logical_or=logical_or, logical_not=logical_not, floormod=_binary(np.mod), add_n=add_n,
)


def _assert_all_finite(x, message="", name=None):
    a = _np(x)
    if not np.all(np.isfinite(a)):
        raise _InvalidArgumentError(message)
    return _wrap(a)


def _assert_equal(x, y, message=None, **kw):
    if not np.all(_np(x) == _np(y)):
        raise _InvalidArgumentError(message or f"assert_equal failed: {x} vs {y}")


debugging = _submodule(
    "debugging", assert_all_finite=_assert_all_finite, assert_equal=_assert_equal,
    assert_rank=lambda *a, **k: None, assert_positive=lambda *a, **k: None, assert_shapes=lambda *a, **k: None,
    assert_greater=lambda *a, **k: None, assert_greater_equal=lambda *a, **k: None, assert_less=lambda *a, **k: None,
    assert_less_equal=lambda *a, **k: None, assert_non_negative=lambda *a, **k: None,
)
errors = _submodule("errors", InvalidArgumentError=_InvalidArgumentError)


def _random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None):  # noqa: A002
    rng = np.random.default_rng(seed)
    return Tensor((rng.standard_normal(_shp(shape)) * stddev + mean).astype(_npdtype(dtype)))


random = _submodule("random", normal=_random_normal, set_seed=lambda *a, **k: None,
                    shuffle=lambda x, **k: Tensor(np.random.default_rng(0).permutation(_np(x))))


def _map_structure(func, *structure, **kw):
    s0 = structure[0]
    if isinstance(s0, (list, tuple)):
        return type(s0)(_map_structure(func, *[s[i] for s in structure]) for i in builtins.range(len(s0)))
    if isinstance(s0, dict):
        return {k: _map_structure(func, *[s[k] for s in structure]) for k in s0}
    return func(*structure)


def _flatten_structure(s):
    if isinstance(s, (list, tuple)):
        return [x for e in s for x in _flatten_structure(e)]
    if isinstance(s, dict):
        return [x for k in sorted(s) for x in _flatten_structure(s[k])]
    return [s]


nest = _submodule("nest", map_structure=_map_structure, flatten=_flatten_structure)
nn = _submodule("nn", softplus=softplus, sigmoid=sigmoid,
                softmax=lambda x, axis=-1, **k: Tensor(_ssp.softmax(_np(x), axis=axis)))
summary = _submodule("summary")
keras = _submodule("keras")
optimizers = _submodule("optimizers")
data = _submodule("data")
io = _submodule("io")
image = _submodule("image")
experimental = _submodule("experimental")
config = _submodule("config")
compat = _submodule("compat")
autograph = _submodule("autograph")
types_ = _submodule("types")
sys.modules[__name__ + ".types"] = types_

# `from tensorflow.python.util.object_identity import Reference` etc. are served by the real sub-packages in this tree.


def __getattr__(name):
    if name.startswith("__") and name.endswith("__"):
        raise AttributeError(name)
    return _DummyBase if name[:1].isupper() else _Dummy()
