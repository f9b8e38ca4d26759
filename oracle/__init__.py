"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of ``oracle/libnmpc_oracle.so`` (plain-C restatement of the reference's NMPC solve path; see
``oracle/nmpc_oracle.h`` for what is pinned and what is "parity unpinned").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.
The product package (``dyobav-mpcnwta-warehouse_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnmpc_oracle.so")

STATUS_NAMES = ("Converged", "NotConvergedIterations", "NotConvergedOutOfTime")


class _Problem(C.Structure):
    _fields_ = [("N", C.c_int32), ("Nother", C.c_int32), ("Nstc", C.c_int32), ("Ndyn", C.c_int32),
                ("ts", C.c_double),
                ("lin_vel_min", C.c_double), ("lin_vel_max", C.c_double), ("ang_vel_max", C.c_double),
                ("lin_acc_min", C.c_double), ("lin_acc_max", C.c_double), ("ang_acc_max", C.c_double),
                ("vehicle_width", C.c_double), ("vehicle_margin", C.c_double), ("social_margin", C.c_double)]


class _Options(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("initial_tolerance", C.c_double), ("delta_tolerance", C.c_double),
                ("max_outer", C.c_int32), ("max_inner", C.c_int32), ("lbfgs_mem", C.c_int32),
                ("initial_penalty", C.c_double), ("penalty_update", C.c_double),
                ("inner_tol_update", C.c_double), ("sufficient_decrease", C.c_double),
                ("lip_delta", C.c_double), ("lip_eps", C.c_double),
                ("cbfgs_alpha", C.c_double), ("cbfgs_eps", C.c_double), ("sy_eps", C.c_double),
                ("akkt_form", C.c_int32), ("hoist_trig", C.c_int32), ("max_time_s", C.c_double),
                ("max_evals", C.c_int32), ("reserved_", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [("cost", C.c_double), ("status", C.c_int32), ("outer_iters", C.c_int32),
                ("inner_iters", C.c_int32), ("n_cost_evals", C.c_int32), ("n_grad_evals", C.c_int32),
                ("last_fpr", C.c_double), ("delta_y_norm", C.c_double), ("f2_norm", C.c_double),
                ("penalty", C.c_double), ("n_points", C.c_int32), ("reserved_", C.c_int32)]


RESULT_DTYPE = np.dtype([("cost", "f8"), ("status", "i4"), ("outer_iters", "i4"), ("inner_iters", "i4"),
                         ("n_cost_evals", "i4"), ("n_grad_evals", "i4"), ("last_fpr", "f8"),
                         ("delta_y_norm", "f8"), ("f2_norm", "f8"), ("penalty", "f8"), ("n_points", "i4"),
                         ("reserved_", "i4")], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(_Result)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``). Building the checker is not using it."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("nmpc_oracle.c", "nmpc_oracle_impl.h", "nmpc_oracle.h"))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_np.restype = C.c_int
        _lib.orc_dist2_to_lineseg.restype = C.c_double
        _lib.orc_dist2_to_lineseg.argtypes = [C.c_double] * 6
        _lib.orc_inside_ellipse.restype = C.c_double
        _lib.orc_inside_ellipse.argtypes = [C.c_double] * 7
        _lib.orc_inside_cvx_polygon.restype = C.c_double
        _lib.orc_inside_cvx_polygon.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


@dataclass
class Problem:
    """Dimensions + robot constants; defaults = config/mpc_default.yaml / mpc_fast.yaml of the reference."""
    N: int = 20
    Nother: int = 10
    Nstc: int = 10
    Ndyn: int = 15
    ts: float = 0.2
    lin_vel_min: float = -0.5
    lin_vel_max: float = 1.5
    ang_vel_max: float = 0.5
    lin_acc_min: float = -1.0
    lin_acc_max: float = 1.0
    ang_acc_max: float = 3.0
    vehicle_width: float = 0.5
    vehicle_margin: float = 0.2
    social_margin: float = 0.2

    def c(self) -> _Problem:
        return _Problem(self.N, self.Nother, self.Nstc, self.Ndyn, self.ts, self.lin_vel_min, self.lin_vel_max,
                        self.ang_vel_max, self.lin_acc_min, self.lin_acc_max, self.ang_acc_max,
                        self.vehicle_width, self.vehicle_margin, self.social_margin)

    @property
    def np_(self) -> int:
        p = self.c()
        return lib().orc_np(C.byref(p))


@dataclass
class Options:
    tolerance: float = 1e-4
    initial_tolerance: float = 1e-4
    delta_tolerance: float = 1e-4
    max_outer: int = 10
    max_inner: int = 500
    lbfgs_mem: int = 10
    initial_penalty: float = 10.0
    penalty_update: float = 5.0
    inner_tol_update: float = 0.1
    sufficient_decrease: float = 0.1
    lip_delta: float = 1e-12
    lip_eps: float = 1e-6
    cbfgs_alpha: float = 1.0
    cbfgs_eps: float = 1e-8
    sy_eps: float = 1e-10
    akkt_form: int = 0   # 0 = OpEn source form of the AKKT residual, 1 = documented form (see nmpc_oracle.h)
    max_time_s: float = 0.0   # wall-clock budget of one solve (the reference's max_solver_time: 0.1 s); 0 = none
    hoist_trig: int = 0       # 1: cos / sin of the ellipse angles once per solve instead of per evaluation (same bits)
    max_evals: int = 0        # evaluation budget of one solve (points, = the kernels' info[4]; nmpc_config.max_evaluations); 0 = none
    extra: dict = field(default_factory=dict)

    def c(self) -> _Options:
        return _Options(self.tolerance, self.initial_tolerance, self.delta_tolerance, self.max_outer,
                        self.max_inner, self.lbfgs_mem, self.initial_penalty, self.penalty_update,
                        self.inner_tol_update, self.sufficient_decrease, self.lip_delta, self.lip_eps,
                        self.cbfgs_alpha, self.cbfgs_eps, self.sy_eps, self.akkt_form, self.hoist_trig, self.max_time_s,
                        self.max_evals, 0)


TRACE_HEAD = 16   # ORC_TRACE_HEAD
TRACE_FIELDS = ("outer", "iter", "lip_doublings", "ls_halvings", "pair", "pairs_held", "gamma", "norm_gfpr", "psi", "tau",
                "n_cost", "n_grad", "margin", "margin_kind", "penalty", "steepness")
MARGIN_KINDS = {0: "-", 1: "lipschitz", 2: "linesearch", 3: "pair", 4: "exit", 5: "outer loop (exit criteria / penalty stall test)"}


def _suffix(dtype, reassoc: bool = False) -> str:
    dtype = np.dtype(dtype)
    if reassoc:
        if dtype != np.float64:
            raise TypeError("the re-associated variant exists in double precision only")
        return "r64"
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(dtype)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def eval_problem(pr: Problem, u, p, dtype=np.float64):
    """f, F1[2N], F2[Ndyn] of the reference's problem definition (mpc_builder.py:45-174)."""
    u = np.ascontiguousarray(u, dtype=dtype)
    p = np.ascontiguousarray(p, dtype=dtype)
    assert u.size == 2 * pr.N and p.size == pr.np_, (u.size, p.size, pr.np_)
    f = np.zeros(1, dtype=dtype)
    F1 = np.zeros(2 * pr.N, dtype=dtype)
    F2 = np.zeros(pr.Ndyn, dtype=dtype)
    cp = pr.c()
    getattr(lib(), "orc_eval_" + _suffix(dtype))(C.byref(cp), _ptr(u), _ptr(p), _ptr(f), _ptr(F1), _ptr(F2))
    return float(f[0]), F1, F2


def psi(pr: Problem, u, c, y, p, grad=True, dtype=np.float64, reassoc=False):
    u = np.ascontiguousarray(u, dtype=dtype)
    p = np.ascontiguousarray(p, dtype=dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    assert u.size == 2 * pr.N and p.size == pr.np_ and y.size == 2 * pr.N
    val = np.zeros(1, dtype=dtype)
    g = np.zeros(2 * pr.N, dtype=dtype) if grad else None
    cp = pr.c()
    cc = C.c_double(c) if np.dtype(dtype) == np.float64 else C.c_float(c)
    getattr(lib(), "orc_psi_" + _suffix(dtype, reassoc))(C.byref(cp), _ptr(u), cc, _ptr(y), _ptr(p), _ptr(val),
                                                 _ptr(g) if grad else None)
    return float(val[0]), g


def solve(pr: Problem, op: Options, p, u0=None, y0=None, dtype=np.float64, reassoc=False):
    """One ALM/PANOC solve. Returns (u, y, result-record). `reassoc`: the same fp64 algorithm with its sums associated
    differently (nmpc_oracle_impl.h, ORC_REASSOC)."""
    p = np.ascontiguousarray(p, dtype=dtype)
    assert p.size == pr.np_
    u = np.zeros(2 * pr.N, dtype=dtype) if u0 is None else np.array(u0, dtype=dtype)
    y = np.zeros(2 * pr.N, dtype=dtype) if y0 is None else np.array(y0, dtype=dtype)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    cp, co = pr.c(), op.c()
    rc = getattr(lib(), "orc_solve_" + _suffix(dtype, reassoc))(C.byref(cp), C.byref(co), _ptr(p), _ptr(u), _ptr(y),
                                                                 _ptr(res))
    if rc != 0:
        raise RuntimeError(f"oracle solve failed rc={rc}")
    return u, y, res[0]


def solve_trace(pr: Problem, op: Options, p, u0=None, y0=None, dtype=np.float64, reassoc=False, max_records=None):
    """One solve with its iteration trace: (u, y, result-record, head[n_rec, 16], U[n_rec, 2N]) -- one record per completed
    inner iteration (fields: TRACE_FIELDS; nmpc_oracle.h), U = the iterate after it."""
    p = np.ascontiguousarray(p, dtype=dtype)
    assert p.size == pr.np_
    n = 2 * pr.N
    u = np.zeros(n, dtype=dtype) if u0 is None else np.array(u0, dtype=dtype)
    y = np.zeros(n, dtype=dtype) if y0 is None else np.array(y0, dtype=dtype)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    max_records = max_records or op.max_outer * (op.max_inner + 1)
    tr = np.zeros((max_records, TRACE_HEAD + n), dtype=np.float64)
    nrec = C.c_int(0)
    cp, co = pr.c(), op.c()
    rc = getattr(lib(), "orc_solve_trace_" + _suffix(dtype, reassoc))(C.byref(cp), C.byref(co), _ptr(p), _ptr(u), _ptr(y),
                                                                       _ptr(res), _ptr(tr), C.c_int(max_records),
                                                                       C.byref(nrec))
    if rc != 0:
        raise RuntimeError(f"oracle solve failed rc={rc}")
    tr = tr[:nrec.value]
    return u, y, res[0], tr[:, :TRACE_HEAD].copy(), tr[:, TRACE_HEAD:].copy()


def solve_batch(pr: Problem, op: Options, P, nthreads: int = 1, dtype=np.float64, reassoc=False):
    """Independent solves (zero initial guess / multipliers), OpenMP over instances."""
    P = np.ascontiguousarray(P, dtype=dtype)
    assert P.ndim == 2 and P.shape[1] == pr.np_
    B = P.shape[0]
    U = np.zeros((B, 2 * pr.N), dtype=dtype)
    res = np.zeros(B, dtype=RESULT_DTYPE)
    cp, co = pr.c(), op.c()
    rc = getattr(lib(), "orc_solve_batch_" + _suffix(dtype, reassoc))(C.byref(cp), C.byref(co), _ptr(P), C.c_int(B),
                                                              _ptr(U), _ptr(res), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle batch solve failed rc={rc}")
    return U, res


# ---- primitives for the known-answer tests --------------------------------------------------------------
def dist2_to_lineseg(px, py, ax, ay, bx, by) -> float:
    return lib().orc_dist2_to_lineseg(px, py, ax, ay, bx, by)


def inside_ellipse(px, py, cx, cy, rx, ry, ang) -> float:
    return lib().orc_inside_ellipse(px, py, cx, cy, rx, ry, ang)


def inside_cvx_polygon(px, py, b, a0, a1) -> float:
    b, a0, a1 = (np.ascontiguousarray(v, dtype=np.float64) for v in (b, a0, a1))
    return lib().orc_inside_cvx_polygon(px, py, _ptr(b), _ptr(a0), _ptr(a1), len(b))


def unicycle_rk4(ts, s, a):
    s = np.ascontiguousarray(s, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_unicycle_rk4(C.c_double(ts), _ptr(s), _ptr(a), _ptr(out))
    return out
