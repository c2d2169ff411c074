"""NumPy stand-in for the part of TensorFlow-Probability that gpflow.Parameter and gpflow.utilities.bijectors use."""
import sys
import types

import numpy as np
import tensorflow as tf

__version__ = "0.19.0"  # the reference's "prod" CI pin


def _np(x):
    return tf._np(x)


class Bijector(tf.Module):
    def __init__(self, validate_args=False, name=None, forward_min_event_ndims=0, **kw):
        super().__init__(name=name or type(self).__name__.lower())
        self._fmen = forward_min_event_ndims

    @property
    def forward_min_event_ndims(self):
        return self._fmen

    def forward(self, x, name=None, **kw):
        return tf.Tensor(self._forward(_np(x)))

    def inverse(self, y, name=None, **kw):
        return tf.Tensor(self._inverse(_np(y)))

    def __call__(self, x):
        return self.forward(x)

    def inverse_log_det_jacobian(self, y, event_ndims=None, name=None, **kw):
        """log |d inverse(y) / dy|, summed over the rightmost `event_ndims` axes (elementwise bijectors only)."""
        v = np.asarray(self._ildj(_np(y)), dtype=np.float64)
        nd = 0 if event_ndims is None else int(event_ndims)
        if nd:
            v = v.sum(axis=tuple(range(v.ndim - nd, v.ndim)))
        return tf.Tensor(v)

    def _ildj(self, y):
        raise NotImplementedError(f"{type(self).__name__}: no inverse_log_det_jacobian in the stand-in")


class Identity(Bijector):
    def __init__(self, validate_args=False, name="identity"):
        super().__init__(name=name)

    def _forward(self, x):
        return x

    def _inverse(self, y):
        return y

    def _ildj(self, y):
        return np.zeros_like(y)


class Softplus(Bijector):
    def __init__(self, hinge_softness=None, low=None, validate_args=False, name="softplus"):
        super().__init__(name=name)
        assert hinge_softness is None
        self.low = low

    def _forward(self, x):
        y = np.logaddexp(0.0, x)
        return y if self.low is None else y + _np(self.low)

    def _inverse(self, y):
        if self.low is not None:
            y = y - _np(self.low)
        return y + np.log(-np.expm1(-y))  # log(exp(y) - 1), the form TFP uses

    def _ildj(self, y):
        if self.low is not None:
            y = y - _np(self.low)
        return -np.log(-np.expm1(-y))  # dx/dy = 1 / (1 - exp(-y))


class Exp(Bijector):
    def __init__(self, validate_args=False, name="exp"):
        super().__init__(name=name)

    def _forward(self, x):
        return np.exp(x)

    def _inverse(self, y):
        return np.log(y)

    def _ildj(self, y):
        return -np.log(y)


class Shift(Bijector):
    def __init__(self, shift, validate_args=False, name="shift"):
        super().__init__(name=name)
        self.shift = shift

    def _forward(self, x):
        return x + _np(self.shift)

    def _inverse(self, y):
        return y - _np(self.shift)

    def _ildj(self, y):
        return np.zeros_like(y)


class Scale(Bijector):
    def __init__(self, scale, validate_args=False, name="scale"):
        super().__init__(name=name)
        self.scale = scale

    def _forward(self, x):
        return x * _np(self.scale)

    def _inverse(self, y):
        return y / _np(self.scale)

    def _ildj(self, y):
        return np.zeros_like(y) - np.log(np.abs(_np(self.scale)))


class Sigmoid(Bijector):
    def __init__(self, low=None, high=None, validate_args=False, name="sigmoid"):
        super().__init__(name=name)
        self.low, self.high = low, high

    def _forward(self, x):
        s = 1.0 / (1.0 + np.exp(-x))
        if self.low is None:
            return s
        lo, hi = _np(self.low), _np(self.high)
        return lo + (hi - lo) * s

    def _inverse(self, y):
        if self.low is not None:
            lo, hi = _np(self.low), _np(self.high)
            y = (y - lo) / (hi - lo)
        return np.log(y) - np.log1p(-y)


class Chain(Bijector):
    """Chain([b1, b2, ...]).forward(x) = b1(b2(...(x)))"""

    def __init__(self, bijectors=None, validate_args=False, name=None):
        self.bijectors = list(bijectors or [])
        super().__init__(name=name or ("chain_of_" + "_of_".join(b.name for b in self.bijectors) if self.bijectors else "identity"))

    def _forward(self, x):
        for b in reversed(self.bijectors):
            x = b._forward(x)
        return x

    def _inverse(self, y):
        for b in self.bijectors:
            y = b._inverse(y)
        return y

    def _ildj(self, y):
        tot = np.zeros_like(y)
        for b in self.bijectors:
            tot = tot + b._ildj(y)
            y = b._inverse(y)
        return tot


class FillTriangular(Bijector):
    """vector [..., n(n+1)/2] <-> lower-triangular [..., n, n].  (The element ORDER differs from TFP's spiral; only the
    round trip forward(inverse(L)) = tril(L) is observable through gpflow.Parameter.)"""

    def __init__(self, upper=False, validate_args=False, name="fill_triangular"):
        super().__init__(name=name, forward_min_event_ndims=1)
        assert not upper

    def _forward(self, x):
        m = x.shape[-1]
        n = int((np.sqrt(8 * m + 1) - 1) / 2)
        out = np.zeros(x.shape[:-1] + (n, n), dtype=x.dtype)
        i, j = np.tril_indices(n)
        out[..., i, j] = x
        return out

    def _inverse(self, y):
        n = y.shape[-1]
        i, j = np.tril_indices(n)
        return y[..., i, j]


class TransformedVariable(tf.Tensor):
    """tfp.util.TransformedVariable: stores pretransformed_input = bijector.inverse(initial_value) in a tf.Variable and
    reads as bijector.forward(variable)."""

    def __init__(self, initial_value, bijector, dtype=None, name=None, trainable=True, shape=None, **kw):
        self._bijector = bijector
        val = tf._np(initial_value)
        if dtype is not None:
            val = val.astype(tf._npdtype(dtype))
        self._pretransformed_input = tf.Variable(bijector._inverse(val), trainable=trainable, name=name)
        self._name = name
        self._shape = None

    def _value(self):
        return np.asarray(self._bijector._forward(self._pretransformed_input._v))

    def _tf_composite_parts(self):
        return [("_pretransformed_input", self._pretransformed_input)]

    @property
    def bijector(self):
        return self._bijector

    @property
    def pretransformed_input(self):
        return self._pretransformed_input

    @property
    def name(self):
        return self._name

    @property
    def trainable_variables(self):
        return (self._pretransformed_input,) if self._pretransformed_input.trainable else ()

    @property
    def variables(self):
        return (self._pretransformed_input,)


class Distribution:
    def __init__(self, *a, **k):
        pass


class _NpDist(Distribution):
    """the few tfp.distributions GPflow's documentation hangs on Parameters as priors: log_prob only"""

    def log_prob(self, value, name=None):
        return tf.Tensor(np.asarray(self._lp(_np(value)), dtype=np.float64))


class Normal(_NpDist):
    def __init__(self, loc, scale, **kw):
        self.loc, self.scale = _np(loc), _np(scale)

    def _lp(self, x):
        z = (x - self.loc) / self.scale
        return -0.5 * z * z - np.log(self.scale) - 0.5 * np.log(2.0 * np.pi)


class Gamma(_NpDist):
    def __init__(self, concentration, rate=None, **kw):
        self.a, self.b = _np(concentration), _np(rate)

    def _lp(self, x):
        from scipy.special import gammaln
        return (self.a - 1.0) * np.log(x) - self.b * x + self.a * np.log(self.b) - gammaln(self.a)


class LogNormal(_NpDist):
    def __init__(self, loc, scale, **kw):
        self.loc, self.scale = _np(loc), _np(scale)

    def _lp(self, x):
        z = (np.log(x) - self.loc) / self.scale
        return -0.5 * z * z - np.log(self.scale * x) - 0.5 * np.log(2.0 * np.pi)


def _mod(name, **attrs):
    m = types.ModuleType(__name__ + "." + name)
    for k, v in attrs.items():
        setattr(m, k, v)

    def _ga(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return tf._DummyBase if attr[:1].isupper() else tf._Dummy()
    m.__getattr__ = _ga
    sys.modules[m.__name__] = m
    return m


bijectors = _mod("bijectors", Bijector=Bijector, Identity=Identity, Softplus=Softplus, Exp=Exp, Shift=Shift, Scale=Scale,
                 Sigmoid=Sigmoid, Chain=Chain, FillTriangular=FillTriangular)
util = _mod("util", TransformedVariable=TransformedVariable)
distributions = _mod("distributions", Distribution=Distribution, Normal=Normal, Gamma=Gamma, LogNormal=LogNormal)
mcmc = _mod("mcmc")
stats = _mod("stats")
math = _mod("math")
