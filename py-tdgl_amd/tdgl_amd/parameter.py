"""``Parameter``: a function of position (and optionally time) with keyword arguments bound, with
the behaviour of the reference's (`tdgl/parameter.py:66-439`): ``func(x, y[, z], **kwargs)`` checked
at construction, ``time_dependent=True`` for functions with a keyword-only ``t``, calls with or
without ``z`` / ``t``, ``+ - * / **`` between parameters and numbers giving ``CompositeParameter``,
equality by code and arguments, pickling, readable ``repr``.  The reference's result cache is not
kept (here a static parameter is evaluated once and lives on the device).  Additions used by the
solver: a factor that does not depend on position is recognised (``uniform_in_space``), so that
``A(t) = f(t) * A_static(r)`` keeps ``A_static`` on the device (``separable_product``).
"""

import inspect
import operator
from numbers import Number
from typing import Callable, Optional, Union

import numpy as np


def function_repr(func: Callable, argspec=None) -> str:
    """``name(arg, kw=default, *, kwonly=default)`` of a function (`tdgl/parameter.py:31-63`);
    ``argspec`` may be any object with the attributes of ``inspect.FullArgSpec``."""
    if argspec is None:
        argspec = inspect.getfullargspec(func)
    get = lambda name: getattr(argspec, name, None)  # noqa: E731
    args = [str(a) for a in (get("args") or [])]
    for k, val in enumerate(list(get("defaults") or [])[::-1]):
        args[-(k + 1)] += f"={val!r}"
    if get("varargs"):
        args.append("*" + get("varargs"))
    if get("kwonlyargs"):
        if not get("varargs"):
            args.append("*")
        args.extend(get("kwonlyargs"))
    for k, name in enumerate(args):
        if get("kwonlydefaults") and name in get("kwonlydefaults"):
            args[k] += f"={get('kwonlydefaults')[name]!r}"
    if get("varkw"):
        args.append("**" + get("varkw"))
    for k, name in enumerate(args):
        if get("annotations") and name in get("annotations"):
            args[k] += f": {getattr(get('annotations')[name], '__name__', get('annotations')[name])!r}"
    return getattr(func, "__name__", "func") + "(" + ", ".join(args) + ")"


class _BoundArgs:
    """argspec stand-in listing a parameter's bound keyword arguments (for ``repr``)."""

    def __init__(self, names, values):
        self.args, self.defaults = list(names), list(values)


def _same(a, b) -> bool:
    if a is b:
        return True
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        return a.shape == b.shape and bool(np.allclose(a, b))
    try:
        return bool(a == b)
    except (TypeError, ValueError):
        return False


class Parameter:
    """``Parameter(func, time_dependent=False, **kwargs)``.

    ``func`` takes ``x, y`` (and optionally ``z`` as third argument) positionally; everything else
    must be a keyword argument (with a default, or keyword-only).  A time-dependent parameter's
    ``func`` takes the time as keyword-only ``t``.  Violations raise ``ValueError`` as in the
    reference (`tdgl/parameter.py:86-128`)."""

    def __init__(self, func: Callable, time_dependent: bool = False, **kwargs):
        kwargs.pop("use_cache", None)  # (reference option; nothing is cached here)
        spec = inspect.getfullargspec(func)
        args = spec.args
        n_pos = 2
        if args[:n_pos] != ["x", "y"]:
            raise ValueError(f"The first function arguments must be x and y, not {', '.join(args[:n_pos])!r}.")
        if "z" in args:
            if args.index("z") != n_pos:
                raise ValueError("If the function takes an argument z, it must be the third argument (x, y, z).")
            n_pos = 3
        defaults = spec.defaults or ()
        if len(defaults) != len(args) - n_pos:
            raise ValueError("All arguments other than x, y, z must be keyword arguments.")
        extra = set(kwargs) - set(args[n_pos:])
        if not extra.issubset(set(spec.kwonlyargs or [])):
            raise ValueError(f"Provided keyword-only arguments ({extra!r}) do not match the function signature: "
                             f"{function_repr(func)}.")
        if time_dependent and "t" not in (spec.kwonlyargs or []):
            raise ValueError("A time-dependent Parameter must take time t as a keyword argument.")
        if "t" in kwargs:
            raise ValueError("'t' cannot be bound as a Parameter keyword argument.")
        self.func = func
        self.time_dependent = bool(time_dependent)
        self.kwargs = dict(zip(args[n_pos:], defaults))
        self.kwargs.update(spec.kwonlydefaults or {})
        self.kwargs.update(kwargs)
        self._takes_z = n_pos == 3
        # True for factors that do not depend on position (e.g. LinearRamp): lets the solver
        # recognise A(t) = f(t) * A_static and keep A_static on the device
        self.uniform_in_space = False
        self.ramp = None  # LinearRamp: dict(tmin, tmax, initial, final)

    def separable_product(self):
        """``(f, static)`` if this parameter is ``f(t) * static(x, y, z)`` with ``f`` uniform in
        space and ``static`` independent of time, else ``None``."""
        return None

    def scalar(self, t) -> float:
        """Value of a uniform-in-space factor at time ``t``."""
        z = np.zeros(1)
        return float(np.ravel(self(z, z, z, t=t))[0])

    def __call__(self, x, y, z=None, t=None):
        """Value at the points (arrays or numbers); the result is squeezed and a 0-d result
        returned as a number (`tdgl/parameter.py:156-172`)."""
        kw = dict(self.kwargs)
        if t is not None:
            kw["t"] = t
        elif self.time_dependent:
            kw["t"] = 0.0
        x, y = np.atleast_1d(x, y)
        if z is not None:
            if not self._takes_z:
                raise TypeError(f"{function_repr(self.func)} does not take z.")
            kw["z"] = np.atleast_1d(z)
        result = np.asarray(self.func(x, y, **kw)).squeeze()
        return result.item() if result.ndim == 0 else result

    def _clear_cache(self):  # reference API; nothing is cached here
        pass

    # -- arithmetic -------------------------------------------------------------------------------
    def __add__(self, other):
        return CompositeParameter(self, other, operator.add)

    def __radd__(self, other):
        return CompositeParameter(other, self, operator.add)

    def __sub__(self, other):
        return CompositeParameter(self, other, operator.sub)

    def __rsub__(self, other):
        return CompositeParameter(other, self, operator.sub)

    def __mul__(self, other):
        return CompositeParameter(self, other, operator.mul)

    def __rmul__(self, other):
        return CompositeParameter(other, self, operator.mul)

    def __truediv__(self, other):
        return CompositeParameter(self, other, operator.truediv)

    def __rtruediv__(self, other):
        return CompositeParameter(other, self, operator.truediv)

    def __pow__(self, other):
        return CompositeParameter(self, other, operator.pow)

    def __rpow__(self, other):
        return CompositeParameter(other, self, operator.pow)

    def __neg__(self):
        return CompositeParameter(-1.0, self, operator.mul)

    # -- identity ---------------------------------------------------------------------------------
    def __eq__(self, other) -> bool:
        """Same code, same bound arguments (`tdgl/parameter.py:246-275`)."""
        if other is self:
            return True
        if not isinstance(other, Parameter) or isinstance(other, CompositeParameter):
            return False
        if getattr(self.func, "__code__", self.func) != getattr(other.func, "__code__", other.func):
            return False
        if set(self.kwargs) != set(other.kwargs):
            return False
        return all(_same(self.kwargs[k], other.kwargs[k]) for k in self.kwargs)

    __hash__ = None

    def _argspec(self):
        names, values = list(self.kwargs), list(self.kwargs.values())
        if self.time_dependent:
            names.insert(0, "time_dependent")
            values.insert(0, True)
        return _BoundArgs(names, values)

    def _bare_repr(self) -> str:
        return function_repr(self.func, self._argspec())

    def __repr__(self):
        return f"{self.__class__.__name__}<{self._bare_repr()}>"

    def __getstate__(self):
        import cloudpickle

        state = self.__dict__.copy()
        state["func"] = cloudpickle.dumps(state["func"])  # (functions defined in a script or a test)
        return state

    def __setstate__(self, state):
        import cloudpickle

        state = dict(state)
        state["func"] = cloudpickle.loads(state["func"])
        self.__dict__.update(state)


class CompositeParameter(Parameter):
    """Result of ``+ - * / **`` between parameters and/or numbers (`tdgl/parameter.py:278-398`)."""

    VALID_OPERATORS = {operator.add: "+", operator.sub: "-", operator.mul: "*", operator.truediv: "/",
                       operator.pow: "**"}

    def __init__(self, left: Union[Number, Parameter], right: Union[Number, Parameter],
                 operator_: Union[Callable, str]):
        valid = (Number, Parameter)
        if not isinstance(left, valid):
            raise TypeError(f"Left must be a number, Parameter, or CompositeParameter, not {type(left)!r}.")
        if not isinstance(right, valid):
            raise TypeError(f"Right must be a number, Parameter, or CompositeParameter, not {type(right)!r}.")
        if isinstance(left, Number) and isinstance(right, Number):
            raise TypeError("Either left or right must be a Parameter or CompositeParameter.")
        if isinstance(operator_, str):
            operator_ = {v: k for k, v in self.VALID_OPERATORS.items()}.get(operator_.strip())
        if operator_ not in self.VALID_OPERATORS:
            raise ValueError(f"Unknown operator, {operator_!r}. Valid operators are {list(self.VALID_OPERATORS)!r}.")
        self.left, self.right, self.operator = left, right, operator_
        self.kwargs = {}
        self.time_dependent = any(isinstance(v, Parameter) and v.time_dependent for v in (left, right))
        self.uniform_in_space = all((not isinstance(v, Parameter)) or v.uniform_in_space for v in (left, right))
        self.ramp = None

    def separable_product(self):
        if self.operator is not operator.mul:
            return None
        for f, static in ((self.left, self.right), (self.right, self.left)):
            if (isinstance(f, Parameter) and f.uniform_in_space and f.time_dependent
                    and isinstance(static, Parameter) and not static.time_dependent):
                return f, static
        return None

    def __call__(self, x, y, z=None, t=None):
        values = []
        for v in (self.left, self.right):
            if isinstance(v, Parameter):
                v = v(x, y, z, t=t) if v.time_dependent else v(x, y, z)
            values.append(v)
        a, b = values
        # a per-point scalar factor next to a vector field scales every component
        if isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.ndim != b.ndim and a.ndim >= 1 and b.ndim >= 1:
            if a.ndim < b.ndim:
                a = a[:, None]
            else:
                b = b[:, None]
        return self.operator(a, b)

    def _bare_repr(self) -> str:
        side = lambda v: v._bare_repr() if isinstance(v, Parameter) else str(v)  # noqa: E731
        return f"({side(self.left)} {self.VALID_OPERATORS[self.operator]} {side(self.right)})"

    def __eq__(self, other) -> bool:
        if other is self:
            return True
        if not isinstance(other, CompositeParameter):
            return False
        return self.left == other.left and self.right == other.right and self.operator is other.operator

    __hash__ = None

    def __getstate__(self):
        return self.__dict__.copy()

    def __setstate__(self, state):
        self.__dict__.update(state)


class Constant(Parameter):
    """A parameter whose value depends on neither position nor time (`tdgl/parameter.py:417-438`)."""

    def __init__(self, value: Number, dimensions: int = 2):
        if dimensions not in (2, 3):
            raise ValueError(f"Dimensions must be 2 or 3, got {dimensions}.")
        if dimensions == 2:
            def constant(x, y, value=0):
                return value * np.ones_like(x)
        else:
            def constant(x, y, z, value=0):
                return value * np.ones_like(x)
        super().__init__(constant, value=value)
        self.uniform_in_space = True


# ---- sources (tdgl/sources/constant.py, scaling.py) ------------------------------------------
def constant_field_vector_potential(x, y, z, *, Bz):
    xs = x - (x.min() + np.ptp(x) / 2)
    ys = y - (y.min() + np.ptp(y) / 2)
    return np.stack([-Bz * ys / 2, Bz * xs / 2, np.zeros_like(xs)], axis=1)


def ConstantField(value: float = 0, field_units: str = "mT", length_units: str = "um") -> Parameter:
    """Vector potential of a uniform out-of-plane field ``value`` (in ``field_units``), in units
    of ``field_units * length_units`` (`tdgl/sources/constant.py:7-39`)."""

    return Parameter(constant_field_vector_potential, Bz=float(value))


def linear_ramp(x, y, z, *, t, tmin, tmax, initial=0.0, final=1.0):
    """``initial`` before ``tmin``, ``final`` after ``tmax``, linear in between -- one number for all
    positions (`tdgl/sources/scaling.py:4-14`)."""
    if t < tmin:
        return initial
    if t < tmax:
        return initial + (final - initial) * (t - tmin) / (tmax - tmin)
    return final


def LinearRamp(tmin: float = 0, tmax: float = 10, initial: float = 0, final: float = 1) -> Parameter:
    """A factor that ramps linearly from ``initial`` to ``final`` between ``tmin`` and ``tmax``
    (`tdgl/sources/scaling.py:17-40`)."""
    p = Parameter(linear_ramp, tmin=tmin, tmax=tmax, initial=initial, final=final, time_dependent=True)
    p.uniform_in_space = True
    p.ramp = dict(tmin=float(tmin), tmax=float(tmax), initial=float(initial), final=float(final))
    return p


def Scale(func, **kwargs) -> Parameter:
    """An arbitrary time-dependent scale factor ``func(x, y, z, *, t, **kwargs)``
    (`tdgl/sources/scaling.py:43-54`)."""
    kwargs["time_dependent"] = True
    return Parameter(func, **kwargs)


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
