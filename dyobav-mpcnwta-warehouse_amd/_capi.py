"""ctypes binding of ``libnmpc_hip.so`` (C ABI declared in ``include/nmpc_hip.h``).

This is the only place the Python host code touches native code. There is deliberately no fallback: if the
library is missing or no HIP device is visible every entry point raises -- results never come from a CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build

EXIT_STATUS_NAMES = ("Converged", "NotConvergedIterations", "NotConvergedOutOfTime", "NotFiniteComputation",
                     "CapacityExceeded", "NotAxisAligned")
ABI_VERSION = 5


class NmpcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libnmpc_hip error {code}: {msg}")
        self.code = code


class NmpcConfigStruct(C.Structure):
    """Mirror of ``struct nmpc_config``."""
    _fields_ = [
        ("abi_version", C.c_int32), ("device_id", C.c_int32),
        ("N_hor", C.c_int32), ("Nother", C.c_int32), ("Nstcobs", C.c_int32), ("Ndynobs", C.c_int32),
        ("ts", C.c_double),
        ("lin_vel_min", C.c_double), ("lin_vel_max", C.c_double), ("ang_vel_max", C.c_double),
        ("lin_acc_min", C.c_double), ("lin_acc_max", C.c_double), ("ang_acc_max", C.c_double),
        ("vehicle_width", C.c_double), ("vehicle_margin", C.c_double), ("social_margin", C.c_double),
        ("tolerance", C.c_double), ("initial_tolerance", C.c_double), ("delta_tolerance", C.c_double),
        ("max_outer_iterations", C.c_int32), ("max_inner_iterations", C.c_int32),
        ("lbfgs_memory", C.c_int32), ("max_active_dynobs", C.c_int32),
        ("initial_penalty", C.c_double), ("penalty_update_factor", C.c_double),
        ("inner_tolerance_update_factor", C.c_double), ("sufficient_decrease_coeff", C.c_double),
        ("lip_eps_f64", C.c_double), ("lip_delta_f64", C.c_double),
        ("lip_eps_f32", C.c_double), ("lip_delta_f32", C.c_double),
        ("cbfgs_alpha", C.c_double), ("cbfgs_epsilon", C.c_double), ("sy_epsilon", C.c_double),
        ("latency_waves", C.c_int32), ("akkt_form", C.c_int32),
        ("max_solver_time_us", C.c_double),
        ("coop_waves", C.c_int32), ("axis_aligned", C.c_int32), ("reg_table", C.c_int32), ("staged", C.c_int32),
        ("polish", C.c_int32), ("polish_max_outer_iterations", C.c_int32), ("polish_max_inner_iterations", C.c_int32),
        ("staged_evals", C.c_int32),
        ("polish_tolerance", C.c_double), ("polish_delta_tolerance", C.c_double),
        ("max_evaluations", C.c_int32), ("tail_latency", C.c_int32), ("batch_invariant", C.c_int32),
    ]


class NmpcLayoutInfo(C.Structure):
    """Mirror of ``struct nmpc_layout_info``."""
    _fields_ = [("np", C.c_int32), ("lds_bytes_f32", C.c_int32), ("lds_bytes_f64", C.c_int32),
                ("reg_slots_f32", C.c_int32), ("global_table_f32", C.c_int32), ("global_table_f64", C.c_int32),
                ("ws_elems_f32", C.c_int64), ("ws_elems_f64", C.c_int64),
                ("table_entries_f32", C.c_int32), ("table_entries_f64", C.c_int32),
                ("dyn_cap", C.c_int32), ("reserved", C.c_int32)]


class NmpcAssembleArgs(C.Structure):
    """Mirror of ``struct nmpc_assemble_args`` (device pointers)."""
    _fields_ = [("last_u", C.c_void_p), ("state", C.c_void_p), ("ref_states", C.c_void_p), ("speed_ref", C.c_void_p),
                ("tuning", C.c_void_p), ("other_robots", C.c_void_p), ("map_polygons", C.c_void_p),
                ("n_map_polygons", C.c_int32), ("n_dyn", C.c_int32), ("dyn_obstacles", C.c_void_p),
                ("stc_weights", C.c_void_p), ("dyn_weights", C.c_void_p), ("selected", C.c_void_p)]


class NmpcLoopArgs(C.Structure):
    """Mirror of ``struct nmpc_loop_args`` (device pointers)."""
    _fields_ = ([(n, C.c_int32) for n in ("B", "n_run", "H", "W", "Lmax", "M", "step", "max_steps")] + [("run", C.c_void_p)] +
                [(n, C.c_double) for n in ("base_speed", "lin_vel_max", "human_size", "human_vmax")] +
                [(n, C.c_void_p) for n in ("robot", "last_u", "humans", "hist", "hcount", "hidx", "hpath", "ref_traj", "ref_len",
                                           "idx_ref", "goal", "polys", "stagger", "alive", "collision", "complete", "steps",
                                           "clr_dyn", "clr_stc", "dev_sum", "dev_max", "n_traj", "traj", "acts", "state_c",
                                           "last_u_c", "refs_c", "speed_c", "dyn_c", "U_c", "y_c", "U", "y")] +
                [("gather_y", C.c_int32), ("n_hyp", C.c_int32)] +
                [(n, C.c_double) for n in ("hyp_fan_rad", "hyp_radius0", "hyp_radius_growth")])


# every symbol include/nmpc_hip.h declares (checked by the CPU test-suite against the built library)
EXPORTED_SYMBOLS = (
    "nmpc_default_config", "nmpc_layout", "nmpc_create", "nmpc_destroy", "nmpc_param_len", "nmpc_set_stream", "nmpc_use_own_stream", "nmpc_set_pointer_mode",
    "nmpc_set_dispatch_order",
    "nmpc_solve_batch_f32", "nmpc_solve_batch_f64", "nmpc_solve_trace_f64", "nmpc_eval_batch_f32", "nmpc_eval_batch_f64",
    "nmpc_assemble_params_f32", "nmpc_assemble_params_f64",
    "nmpc_hypotheses_to_ellipses_f32", "nmpc_hypotheses_to_ellipses_f64",
    "nmpc_loop_pre_f32", "nmpc_loop_pre_f64", "nmpc_loop_post_f32", "nmpc_loop_post_f64",
    "nmpc_last_kernel_ms", "nmpc_last_launch_info", "nmpc_kernel_info", "nmpc_selftest", "nmpc_last_error",
)

_lib: Optional[C.CDLL] = None


def library_path() -> str:
    return _build.LIB_PATH


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the in-tree ``libnmpc_hip.so`` (building it with hipcc first if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64. If this library
    # pulled in /opt/rocm's copies first, a later `import torch` would find "No HIP GPUs". Loading torch first
    # makes the dynamic linker resolve our NEEDED libamdhip64.so.* to the copy that is already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise FileNotFoundError(f"{path} not built; run dyobav-mpcnwta-warehouse_amd/build.py")
        _build.build()
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    lib.nmpc_last_error.restype = C.c_char_p
    lib.nmpc_last_error.argtypes = []
    lib.nmpc_default_config.argtypes = [C.POINTER(NmpcConfigStruct)]
    lib.nmpc_layout.argtypes = [C.POINTER(NmpcConfigStruct), C.POINTER(NmpcLayoutInfo)]
    lib.nmpc_create.argtypes = [C.POINTER(NmpcConfigStruct), C.POINTER(vp)]
    lib.nmpc_destroy.argtypes = [vp]
    lib.nmpc_param_len.argtypes = [vp]
    lib.nmpc_set_stream.argtypes = [vp, vp]
    lib.nmpc_use_own_stream.argtypes = [vp]
    lib.nmpc_set_pointer_mode.argtypes = [vp, i32]
    lib.nmpc_set_dispatch_order.argtypes = [vp, vp, i32]
    for sfx in ("f32", "f64"):
        getattr(lib, "nmpc_solve_batch_" + sfx).argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32]
        getattr(lib, "nmpc_eval_batch_" + sfx).argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp]
        getattr(lib, "nmpc_assemble_params_" + sfx).argtypes = [vp, C.POINTER(NmpcAssembleArgs), i32, vp]
        getattr(lib, "nmpc_hypotheses_to_ellipses_" + sfx).argtypes = [vp, vp, i32, vp, i32, C.c_double, C.c_double,
                                                                       C.c_double, C.c_double, i32, vp, vp]
        getattr(lib, "nmpc_loop_pre_" + sfx).argtypes = [vp, C.POINTER(NmpcLoopArgs)]
        getattr(lib, "nmpc_loop_post_" + sfx).argtypes = [vp, C.POINTER(NmpcLoopArgs)]
    lib.nmpc_solve_trace_f64.argtypes = [vp, vp, vp, vp, C.c_double, vp, vp, vp, vp, vp, vp, i32, C.POINTER(i32)]
    lib.nmpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.nmpc_kernel_info.argtypes = [vp] + [C.POINTER(i32)] * 5
    lib.nmpc_last_launch_info.argtypes = [vp, C.POINTER(i32 * 8)]
    lib.nmpc_selftest.argtypes = [vp]
    for name in EXPORTED_SYMBOLS:
        if name != "nmpc_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _check(rc: int) -> int:
    if rc < 0:
        raise NmpcError(rc, load_library().nmpc_last_error().decode(errors="replace"))
    return rc


def default_config_struct() -> NmpcConfigStruct:
    cfg = NmpcConfigStruct()
    _check(load_library().nmpc_default_config(C.byref(cfg)))
    return cfg


def layout_info(cfg: NmpcConfigStruct) -> NmpcLayoutInfo:
    """``nmpc_layout``: the dimension bookkeeping of a configuration (no device needed)."""
    out = NmpcLayoutInfo()
    _check(load_library().nmpc_layout(C.byref(cfg), C.byref(out)))
    return out


def _suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"unsupported dtype {dtype}; the kernels compute in float32 or float64")


class _Arg:
    """A host numpy array or a raw device pointer (``int``) to hand to the C ABI."""

    @staticmethod
    def ptr(x) -> Optional[int]:
        if x is None:
            return None
        if isinstance(x, np.ndarray):
            assert x.flags.c_contiguous
            return x.ctypes.data
        if isinstance(x, int):
            return x
        if hasattr(x, "data_ptr"):   # torch tensor (host or device); must be contiguous
            assert x.is_contiguous()
            return x.data_ptr()
        raise TypeError(type(x))


class Handle:
    """RAII wrapper of ``nmpc_handle``: one HIP device + stream + workspace."""

    def __init__(self, cfg: NmpcConfigStruct):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.cfg = cfg
        _check(self._lib.nmpc_create(C.byref(cfg), C.byref(self._h)))
        self.np_ = _check(self._lib.nmpc_param_len(self._h))
        self.n = 2 * cfg.N_hor

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------------
    def set_stream(self, stream_ptr: Optional[int]):
        """``stream_ptr``: a ``hipStream_t`` as an int (0 = the null stream = torch's default stream); ``None`` = the
        handle's own non-blocking stream."""
        if stream_ptr is None:
            _check(self._lib.nmpc_use_own_stream(self._h))
        else:
            _check(self._lib.nmpc_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def set_pointer_mode(self, mode: int):
        """0 = classify every array argument per call (default), 1 = all host pointers, 2 = all device pointers."""
        _check(self._lib.nmpc_set_pointer_mode(self._h, int(mode)))

    def set_dispatch_order(self, order=None):
        """Workgroup b of the following solves of ``len(order)`` instances takes instance ``order[b]`` (a permutation:
        numpy int32 array, or an int32 device tensor on the handle's stream; e.g. ``np.argsort(-previous_evals)`` =
        longest first). The library copies it: a host array before the call returns, a device tensor asynchronously on
        the handle's stream -- so a tensor produced on another stream must be kept alive (and that stream synchronised)
        by the caller; with ``set_stream(torch's current stream)`` the copy is ordered like any other torch operation.
        ``None`` clears it."""
        if order is None:
            _check(self._lib.nmpc_set_dispatch_order(self._h, None, 0))
        elif isinstance(order, np.ndarray) or not hasattr(order, "data_ptr"):
            a = np.ascontiguousarray(order, dtype=np.int32)
            _check(self._lib.nmpc_set_dispatch_order(self._h, C.c_void_p(a.ctypes.data), int(a.size)))
        else:
            assert str(order.dtype) == "torch.int32" and order.is_contiguous()
            _check(self._lib.nmpc_set_dispatch_order(self._h, C.c_void_p(order.data_ptr()), int(order.numel())))

    def selftest(self) -> int:
        return _check(self._lib.nmpc_selftest(self._h))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(self._lib.nmpc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_launch_info(self) -> dict:
        """Which kernel family / variant the last solve or eval call launched (``nmpc_last_launch_info``)."""
        v = (C.c_int32 * 8)()
        _check(self._lib.nmpc_last_launch_info(self._h, C.byref(v)))
        return {"family": ("throughput", "latency", "cooperative")[v[0]], "axis_aligned": int(v[1]),
                "staged_outer_iterations": int(v[2]), "polish_selected": int(v[3]), "tail_handed_off": int(v[4])}

    def kernel_info(self) -> dict:
        v = [C.c_int32() for _ in range(5)]
        _check(self._lib.nmpc_kernel_info(self._h, *[C.byref(x) for x in v]))
        keys = ("lds_bytes_f32", "lds_bytes_f64", "lanes_per_step", "waves_per_cu_f32", "waves_per_cu_f64")
        return dict(zip(keys, (int(x.value) for x in v)))

    def solve_raw(self, dtype, P, B, U, cost=None, status=None, iters=None, u0=None, y=None, y_is_input=False,
                  c0=None, info=None, sync=True):
        """Thin call of ``nmpc_solve_batch_*``; every array argument may be numpy, torch or a raw pointer."""
        fn = getattr(self._lib, "nmpc_solve_batch_" + _suffix(dtype))
        p = _Arg.ptr
        _check(fn(self._h, p(P), int(B), p(U), p(cost), p(status), p(iters), p(u0), p(y), int(bool(y_is_input)),
                  p(c0), p(info), int(bool(sync))))

    def assemble_params(self, dtype, B, P_out, last_u, state, ref_states, speed_ref, tuning, stc_weights, dyn_weights,
                        map_polygons=None, dyn_obstacles=None, other_robots=None, selected=None):
        """``nmpc_assemble_params_*``: device tensors in (torch / raw pointers), ``P_out[B, np]`` written on the
        handle's stream. ``map_polygons`` [M,4,2]; ``dyn_obstacles`` [B,n_dyn,N+1,6]."""
        a = NmpcAssembleArgs()
        p = _Arg.ptr
        a.last_u, a.state, a.ref_states, a.speed_ref = p(last_u), p(state), p(ref_states), p(speed_ref)
        a.tuning, a.stc_weights, a.dyn_weights = p(tuning), p(stc_weights), p(dyn_weights)
        a.other_robots = p(other_robots)
        a.map_polygons = p(map_polygons)
        a.n_map_polygons = 0 if map_polygons is None else int(map_polygons.shape[0])
        a.dyn_obstacles = p(dyn_obstacles)
        a.n_dyn = 0 if dyn_obstacles is None else int(dyn_obstacles.shape[1])
        a.selected = p(selected)
        fn = getattr(self._lib, "nmpc_assemble_params_" + _suffix(dtype))
        _check(fn(self._h, C.byref(a), int(B), p(P_out)))

    def loop_step(self, dtype, args: "NmpcLoopArgs", post: bool):
        """``nmpc_loop_pre_*`` / ``nmpc_loop_post_*``: one half of a closed-loop time step (row f3), enqueued on the
        handle's stream."""
        fn = getattr(self._lib, ("nmpc_loop_post_" if post else "nmpc_loop_pre_") + _suffix(dtype))
        _check(fn(self._h, C.byref(args)))

    def hypotheses_to_ellipses(self, dtype, hypos, cur, dyn_out, n_obs_out=None, human_size=0.2, eps=1.0, enlarge=2.0,
                               extra_margin=0.0):
        """``nmpc_hypotheses_to_ellipses_*``: device tensors ``hypos[B,N,P,2]``, ``cur[B,H,2]`` ->
        ``dyn_out[B,Ndynobs,N+1,6]`` (+ ``n_obs_out[B]`` int32)."""
        B, P, H = int(hypos.shape[0]), int(hypos.shape[2]), int(cur.shape[1])
        fn = getattr(self._lib, "nmpc_hypotheses_to_ellipses_" + _suffix(dtype))
        p = _Arg.ptr
        _check(fn(self._h, p(hypos), P, p(cur), H, float(human_size), float(eps), float(enlarge), float(extra_margin),
                  B, p(dyn_out), p(n_obs_out)))

    def solve(self, P: np.ndarray, u0=None, y0=None, c0=None, dtype=None, want_info=True) -> dict:
        """Solve a batch held in host memory; returns numpy arrays."""
        dtype = np.dtype(dtype or P.dtype)
        P = np.ascontiguousarray(P, dtype=dtype)
        if P.ndim != 2 or P.shape[1] != self.np_:
            raise ValueError(f"P must be [B, {self.np_}], got {P.shape}")
        B = P.shape[0]
        U = np.empty((B, self.n), dtype=dtype)
        cost = np.empty(B, dtype=dtype)
        status = np.empty(B, dtype=np.int32)
        iters = np.empty((B, 2), dtype=np.int32)
        y = np.zeros((B, self.n), dtype=dtype) if y0 is None else np.ascontiguousarray(y0, dtype=dtype).copy()
        u0 = None if u0 is None else np.ascontiguousarray(u0, dtype=dtype)
        c0 = None if c0 is None else np.ascontiguousarray(c0, dtype=dtype)
        info = np.empty((B, 8), dtype=dtype) if want_info else None
        self.solve_raw(dtype, P, B, U, cost, status, iters, u0, y, y0 is not None, c0, info, True)
        return dict(U=U, cost=cost, status=status, iters=iters, y=y, info=info)

    def solve_trace(self, p, u0=None, y0=None, c0=None, max_records=None) -> dict:
        """``nmpc_solve_trace_f64``: one instance through the one-wavefront fp64 kernel with its iteration trace --
        ``head[n_rec, 16]`` (fields as in ``include/nmpc_hip.h``) and ``Ut[n_rec, 2N]``, the iterate after each inner
        iteration. Diagnostic (first-divergence audit against the oracle's trace)."""
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        if p.size != self.np_:
            raise ValueError(f"p must have {self.np_} entries, got {p.size}")
        n = self.n
        max_records = int(max_records or self.cfg.max_outer_iterations * (self.cfg.max_inner_iterations + 1))
        u0 = None if u0 is None else np.ascontiguousarray(u0, dtype=np.float64).reshape(n)
        y0 = None if y0 is None else np.ascontiguousarray(y0, dtype=np.float64).reshape(n)
        U, y = np.empty(n), np.empty(n)
        status, iters, info = np.empty(1, np.int32), np.empty(2, np.int32), np.empty(8)
        tr = np.zeros((max_records, 16 + n))
        nrec = C.c_int32(0)
        q = _Arg.ptr
        _check(self._lib.nmpc_solve_trace_f64(self._h, q(p), q(u0), q(y0), float(c0 or 0.0), q(U), q(y), q(status), q(iters),
                                              q(info), q(tr), max_records, C.byref(nrec)))
        tr = tr[:nrec.value]
        return dict(U=U, y=y, status=int(status[0]), iters=iters, info=info, head=tr[:, :16].copy(), Ut=tr[:, 16:].copy())

    def eval(self, P: np.ndarray, U: np.ndarray, Y: np.ndarray, Cpen: np.ndarray, grad=True, dtype=None) -> dict:
        dtype = np.dtype(dtype or P.dtype)
        P = np.ascontiguousarray(P, dtype=dtype)
        U = np.ascontiguousarray(U, dtype=dtype)
        Y = np.ascontiguousarray(Y, dtype=dtype)
        Cpen = np.ascontiguousarray(Cpen, dtype=dtype)
        B = P.shape[0]
        assert P.shape == (B, self.np_) and U.shape == (B, self.n) and Y.shape == (B, self.n) and Cpen.shape == (B,)
        psi = np.empty(B, dtype=dtype)
        g = np.empty((B, self.n), dtype=dtype) if grad else None
        f2 = np.empty(B, dtype=dtype)
        fn = getattr(self._lib, "nmpc_eval_batch_" + _suffix(dtype))
        p = _Arg.ptr
        _check(fn(self._h, p(P), p(U), p(Y), p(Cpen), B, p(psi), p(g), p(f2)))
        return dict(psi=psi, grad=g, f2sq=f2)
