"""``Parameter``: a function of position (and optionally time) with keyword arguments bound, as
in the reference (`tdgl/parameter.py:66-439`), reduced to what the solver path needs: calling,
the ``time_dependent`` flag, and arithmetic between parameters / numbers (the docs' flagship
example is ``LinearRamp(...) * ConstantField(...)``).  The reference's result caching is omitted.
"""

import inspect
import operator
from typing import Callable, Union

import numpy as np


def _takes_time(func: Callable) -> bool:
    return "t" in inspect.getfullargspec(func).kwonlyargs


class Parameter:
    """``Parameter(func, **kwargs)``: ``func(x, y, z, **kwargs)`` or, if ``func`` has a
    keyword-only argument ``t``, ``func(x, y, z, *, t, **kwargs)`` (then ``time_dependent``)."""

    def __init__(self, func: Callable, **kwargs):
        self.func = func
        self.kwargs = kwargs
        self.time_dependent = _takes_time(func)
        # True for factors that do not depend on position (e.g. LinearRamp): lets the solver
        # recognise A(t) = f(t) * A_static and keep A_static on the device
        self.uniform_in_space = False
        self.ramp = None  # LinearRamp: dict(tmin, tmax, initial, final)
        if "t" in kwargs:
            raise ValueError("'t' cannot be bound as a Parameter keyword argument.")

    def separable_product(self):
        """``(f, static)`` if this parameter is ``f(t) * static(x, y, z)`` with ``f`` uniform in
        space and ``static`` independent of time, else ``None``."""
        return None

    def scalar(self, t) -> float:
        """Value of a uniform-in-space factor at time ``t``."""
        z = np.zeros(1)
        return float(np.ravel(self(z, z, z, t=t))[0])

    def __call__(self, x, y, z, t=None):
        kw = dict(self.kwargs)
        if self.time_dependent:
            kw["t"] = 0.0 if t is None else t
        return np.asarray(self.func(np.asarray(x), np.asarray(y), np.asarray(z), **kw))

    def _combine(self, other, op, reflected=False):
        return CompositeParameter(other, self, op) if reflected else CompositeParameter(self, other, op)

    def __add__(self, other):
        return self._combine(other, operator.add)

    def __radd__(self, other):
        return self._combine(other, operator.add, True)

    def __sub__(self, other):
        return self._combine(other, operator.sub)

    def __rsub__(self, other):
        return self._combine(other, operator.sub, True)

    def __mul__(self, other):
        return self._combine(other, operator.mul)

    def __rmul__(self, other):
        return self._combine(other, operator.mul, True)

    def __truediv__(self, other):
        return self._combine(other, operator.truediv)

    def __rtruediv__(self, other):
        return self._combine(other, operator.truediv, True)

    def __neg__(self):
        return self._combine(-1.0, operator.mul)

    def _clear_cache(self):  # reference API; nothing is cached here
        pass

    def __repr__(self):
        args = ", ".join(f"{k}={v!r}" for k, v in self.kwargs.items())
        return f"Parameter<{getattr(self.func, '__name__', 'func')}({args})>"


class CompositeParameter(Parameter):
    """Result of arithmetic between parameters and/or numbers."""

    def __init__(self, left: Union[Parameter, float], right: Union[Parameter, float], op: Callable):
        self.left, self.right, self.op = left, right, op
        self.kwargs = {}
        self.time_dependent = bool(getattr(left, "time_dependent", False) or getattr(right, "time_dependent", False))
        uniform = [(not isinstance(v, Parameter)) or v.uniform_in_space for v in (left, right)]
        self.uniform_in_space = all(uniform)
        self.ramp = None

    def separable_product(self):
        if self.op is not operator.mul:
            return None
        for f, static in ((self.left, self.right), (self.right, self.left)):
            if (isinstance(f, Parameter) and f.uniform_in_space and f.time_dependent
                    and isinstance(static, Parameter) and not static.time_dependent):
                return f, static
        return None

    def __call__(self, x, y, z, t=None):
        def ev(v):
            return v(x, y, z, t=t) if isinstance(v, Parameter) else v

        a, b = ev(self.left), ev(self.right)
        # a scalar-in-space factor (e.g. a ramp returning shape (n,)) scales every component
        if isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.ndim != b.ndim:
            if a.ndim == 1:
                a = a[:, None]
            else:
                b = b[:, None]
        return self.op(a, b)

    def __repr__(self):
        return f"CompositeParameter<{self.left!r} {self.op.__name__} {self.right!r}>"


# ---- sources (tdgl/sources/constant.py, scaling.py) ------------------------------------------
def ConstantField(value: float = 0, field_units: str = "mT", length_units: str = "um") -> Parameter:
    """Vector potential of a uniform out-of-plane field ``value`` (in ``field_units``), in units
    of ``field_units * length_units`` (`tdgl/sources/constant.py:7-39`)."""

    def constant_field_vector_potential(x, y, z, *, Bz):
        xs = x - (x.min() + np.ptp(x) / 2)
        ys = y - (y.min() + np.ptp(y) / 2)
        return np.stack([-Bz * ys / 2, Bz * xs / 2, np.zeros_like(xs)], axis=1)

    return Parameter(constant_field_vector_potential, Bz=float(value))


def LinearRamp(tmin: float = 0, tmax: float = 10, initial: float = 0, final: float = 1) -> Parameter:
    """A factor that ramps linearly from ``initial`` to ``final`` between ``tmin`` and ``tmax``
    (`tdgl/sources/scaling.py`)."""

    def linear_ramp(x, y, z, *, t, tmin, tmax, initial, final):
        frac = np.clip((t - tmin) / (tmax - tmin), 0.0, 1.0)
        return (initial + (final - initial) * frac) * np.ones_like(x, dtype=float)

    p = Parameter(linear_ramp, tmin=tmin, tmax=tmax, initial=initial, final=final)
    p.uniform_in_space = True
    p.ramp = dict(tmin=float(tmin), tmax=float(tmax), initial=float(initial), final=float(final))
    return p


# ---- tabulated time dependence, evaluated on the device ------------------------------------------
class PiecewiseLinear:
    """``f(t)``: linear between the nodes ``(times[k], values[k])``, constant outside.  Time-dependent
    inputs given in this form are uploaded once and evaluated by the time loop itself
    (`tdgl_set_mu_boundary_table` / `tdgl_set_epsilon_table`): no Python call, no upload per step."""

    def __init__(self, times, values):
        self.times = np.asarray(times, dtype=float)
        self.values = np.asarray(values, dtype=float)
        if self.times.ndim != 1 or self.times.shape != self.values.shape or len(self.times) < 1:
            raise ValueError("times and values must be one-dimensional and of equal length")
        if np.any(np.diff(self.times) <= 0):
            raise ValueError("times must increase strictly")

    def __call__(self, t) -> float:
        return float(np.interp(t, self.times, self.values))


class TabulatedCurrents:
    """``terminal_currents`` as tables: ``TabulatedCurrents(times, dict(source=[...], drain=[...]))``.
    Callable like any ``t -> {name: current}`` function (so it works wherever the reference's
    callable does); the solver uploads the tables and the device evaluates them per step."""

    def __init__(self, times, currents):
        self.times = np.asarray(times, dtype=float)
        self.tables = {name: PiecewiseLinear(self.times, v) for name, v in currents.items()}

    def __call__(self, t):
        return {name: f(t) for name, f in self.tables.items()}

    def scaled(self, factor: float) -> "TabulatedCurrents":
        return TabulatedCurrents(self.times, {name: factor * f.values for name, f in self.tables.items()})


class SeparableEpsilon:
    """``disorder_epsilon(r, t) = factor(t) * static(r)``: ``static`` an array over the sites or a
    callable ``static(r[n, 2]) -> [n]``, ``factor`` a `PiecewiseLinear`.  Has the reference's calling
    convention (``epsilon(r, *, t, vectorized=True)``, `tdgl/solver/solver.py:191-216`); the solver
    keeps ``static`` on the device and evaluates ``factor`` inside the time loop."""

    def __init__(self, static, factor: PiecewiseLinear):
        self.static, self.factor = static, factor

    def static_values(self, sites) -> np.ndarray:
        s = self.static(np.asarray(sites)) if callable(self.static) else self.static
        return np.asarray(s, dtype=float) * np.ones(len(sites))

    def __call__(self, r, *, t=0.0, vectorized=True):
        return self.factor(t) * self.static_values(np.atleast_2d(r))
