"""Numeric stand-in for the tiny CasADi / opengen surface the reference's problem definition touches.

TEST-FIXTURE TOOLING (authoring container only). ``casadi`` and ``opengen`` are not installable here, so the
reference's own Python files (``/root/reference/src/pkg_mpc_tracker/solver_build/{mpc_builder,mpc_cost,
mpc_helper}.py`` and ``basic_motion_model/motion_model.py``) are imported with ``casadi.casadi`` and ``opengen``
replaced by the modules built below and *evaluated numerically*: ``SX.sym(name, n)`` returns the numbers
registered under ``name`` instead of a symbol, so running ``MpcModule.build(..., test=True)`` computes
f(u, p), F1(u, p), F2(u, p) for those numbers with the reference's own code. Nothing in here restates the
reference's maths; it only provides matrix plumbing with CasADi's conventions:

* every value is a dense 2-D matrix; ``M[i]`` / ``M[a:b:c]`` / ``M[[...]]`` index the column-major flattening
  and keep the orientation of a row vector (CasADi semantics);
* ``M[r, c]`` never drops a dimension;
* ``reshape`` is column-major; ``sum1`` sums over rows (one value per column), ``sum2`` over columns;
* element-wise binary operations broadcast a scalar or a matching vector.
"""
from __future__ import annotations

import sys
import types

import numpy as np

SYM_VALUES: dict[str, np.ndarray] = {}     # name -> numbers handed out by SX.sym(name, ...)
CAPTURED: dict[str, object] = {}           # filled by the opengen stand-in


def _arr(x) -> np.ndarray:
    if isinstance(x, SX):
        return x.a
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


class SX:
    __array_priority__ = 1000

    def __init__(self, x=0.0):
        self.a = np.array(_arr(x), dtype=np.float64)

    # ---- construction -------------------------------------------------------------------------------
    @staticmethod
    def sym(name, n=1, m=1):
        v = np.asarray(SYM_VALUES[name], dtype=np.float64)
        assert v.size == n * m, (name, v.size, n, m)
        return SX(v.reshape((n, m), order="F"))

    @staticmethod
    def ones(n=1, m=1):
        return SX(np.ones((n, m)))

    @staticmethod
    def zeros(n=1, m=1):
        return SX(np.zeros((n, m)))

    # ---- shape --------------------------------------------------------------------------------------
    @property
    def shape(self):
        return self.a.shape

    @property
    def T(self):
        return SX(self.a.T)

    def size1(self):
        return self.a.shape[0]

    def size2(self):
        return self.a.shape[1]

    def __float__(self):
        assert self.a.size == 1
        return float(self.a.flat[0])

    def __repr__(self):
        return f"SX({self.a!r})"

    # ---- indexing -----------------------------------------------------------------------------------
    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            r, c = idx
            r = [r] if isinstance(r, (int, np.integer)) else r
            c = [c] if isinstance(c, (int, np.integer)) else c
            rows = np.arange(self.a.shape[0])[r]
            cols = np.arange(self.a.shape[1])[c]
            return SX(self.a[np.ix_(np.atleast_1d(rows), np.atleast_1d(cols))])
        flat = self.a.flatten(order="F")
        if isinstance(idx, (int, np.integer)):
            return SX(flat[idx])
        sel = flat[idx] if isinstance(idx, slice) else flat[list(idx)]
        if self.a.shape[0] == 1 and self.a.shape[1] != 1:
            return SX(sel.reshape(1, -1))
        return SX(sel.reshape(-1, 1))

    # ---- arithmetic ---------------------------------------------------------------------------------
    def _bin(self, other, f, swap=False):
        x, y = self.a, _arr(other)
        if swap:
            x, y = y, x
        return SX(f(x, y))

    def __add__(self, o):
        return self._bin(o, np.add)

    def __radd__(self, o):
        return self._bin(o, np.add, True)

    def __sub__(self, o):
        return self._bin(o, np.subtract)

    def __rsub__(self, o):
        return self._bin(o, np.subtract, True)

    def __mul__(self, o):
        return self._bin(o, np.multiply)

    def __rmul__(self, o):
        return self._bin(o, np.multiply, True)

    def __truediv__(self, o):
        return self._bin(o, np.divide)

    def __rtruediv__(self, o):
        return self._bin(o, np.divide, True)

    def __pow__(self, o):
        return self._bin(o, np.power)

    def __neg__(self):
        return SX(-self.a)

    def __lt__(self, o):
        return SX((self.a < _arr(o)).astype(np.float64))

    def __gt__(self, o):
        return SX((self.a > _arr(o)).astype(np.float64))


def _un(f):
    return lambda x: SX(f(_arr(x)))


def _make_casadi_module() -> types.ModuleType:
    m = types.ModuleType("casadi.casadi")
    m.SX = SX
    m.MX = type("MX", (), {})
    m.pi = np.pi
    m.vertcat = lambda *xs: SX(np.vstack([_arr(x) for x in xs]))
    m.horzcat = lambda *xs: SX(np.hstack([_arr(x) for x in xs]))
    m.vcat = lambda xs: m.vertcat(*xs)
    m.hcat = lambda xs: m.horzcat(*xs)
    m.transpose = lambda x: SX(_arr(x).T)
    m.reshape = lambda x, shp: SX(_arr(x).reshape(tuple(shp), order="F"))
    m.fmax = lambda x, y: SX(np.maximum(_arr(x), _arr(y)))
    m.fmin = lambda x, y: SX(np.minimum(_arr(x), _arr(y)))
    m.sum1 = lambda x: SX(_arr(x).sum(axis=0, keepdims=True))
    m.sum2 = lambda x: SX(_arr(x).sum(axis=1, keepdims=True))
    m.dot = lambda x, y: SX(np.sum(_arr(x) * _arr(y)))
    m.mtimes = lambda x, y: SX(_arr(x) @ _arr(y))
    m.norm_2 = lambda x: SX(np.sqrt(np.sum(_arr(x) ** 2)))
    m.mmin = lambda x: SX(np.min(_arr(x)))
    m.mmax = lambda x: SX(np.max(_arr(x)))
    m.sqrt = _un(np.sqrt)
    m.cos = _un(np.cos)
    m.sin = _un(np.sin)
    m.acos = _un(np.arccos)
    m.sign = _un(np.sign)
    m.fabs = _un(np.abs)
    return m


class _Chain:
    """Builder objects of opengen: every ``with_*`` call is recorded and returns ``self``."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs, self.calls = args, kwargs, []

    def __getattr__(self, name):
        if name.startswith("with_"):
            def rec(*a, **k):
                self.calls.append((name, a, k))
                return self
            return rec
        raise AttributeError(name)


class _Rectangle:
    def __init__(self, xmin, xmax):
        self.xmin, self.xmax = list(xmin), list(xmax)


class _Problem(_Chain):
    def __init__(self, u, p, cost):
        super().__init__(u, p, cost)
        CAPTURED.clear()
        CAPTURED.update(u=u, p=p, cost=cost)

    def with_constraints(self, c):
        CAPTURED["bounds"] = c
        return self

    def with_aug_lagrangian_constraints(self, f1, set_c, set_y=None):
        CAPTURED["F1"], CAPTURED["set_c"] = f1, set_c
        return self

    def with_penalty_constraints(self, f2):
        CAPTURED["F2"] = f2
        return self


class _OptimizerBuilder(_Chain):
    def __init__(self, problem, meta, build_config, solver_config):
        super().__init__(problem, meta, build_config, solver_config)
        CAPTURED["solver_config_calls"] = list(solver_config.calls)
        CAPTURED["meta_calls"] = list(meta.calls)
        CAPTURED["build_config_calls"] = list(build_config.calls)

    def build(self):
        raise RuntimeError("the numeric stand-in cannot compile a solver (no cargo / opengen here)")


def _make_opengen_module() -> types.ModuleType:
    og = types.ModuleType("opengen.opengen")
    og.constraints = types.SimpleNamespace(Rectangle=_Rectangle)
    og.builder = types.SimpleNamespace(Problem=_Problem, OpEnOptimizerBuilder=_OptimizerBuilder)
    og.config = types.SimpleNamespace(BuildConfiguration=_Chain, OptimizerMeta=_Chain,
                                      SolverConfiguration=_Chain)
    og.tcp = types.SimpleNamespace(solver_status=types.SimpleNamespace(SolverStatus=object),
                                   OptimizerTcpManager=object)
    return og


def install():
    """Register the stand-ins as ``casadi``, ``casadi.casadi``, ``opengen`` and ``opengen.opengen``."""
    cs = _make_casadi_module()
    pkg = types.ModuleType("casadi")
    pkg.casadi = cs
    pkg.__path__ = []
    for k, v in cs.__dict__.items():
        if not k.startswith("__"):
            setattr(pkg, k, v)
    sys.modules["casadi"] = pkg
    sys.modules["casadi.casadi"] = cs
    og = _make_opengen_module()
    ogpkg = types.ModuleType("opengen")
    ogpkg.opengen = og
    ogpkg.__path__ = []
    for k, v in og.__dict__.items():
        if not k.startswith("__"):
            setattr(ogpkg, k, v)
    sys.modules["opengen"] = ogpkg
    sys.modules["opengen.opengen"] = og
    return cs, og
