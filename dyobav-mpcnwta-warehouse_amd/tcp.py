"""TCP/JSON front end of the solver (row f4): the wire format OpEn's generated ``tcp_iface`` server speaks and
``opengen.tcp.OptimizerTcpManager`` expects, so that ``TrajectoryTracker(use_tcp=True)``
(``/root/reference/src/pkg_mpc_tracker/trajectory_tracker.py:62-66, 385-400``) and non-Python callers keep working.

opengen is not vendored in the reference; the protocol below is restated from its published behaviour (opengen 0.6.x,
``opengen/tcp/optimizer_tcp_manager.py`` and the generated ``tcp_iface/src/main.rs``) -- **wire format unpinned**, like
the solver algorithm. One request per connection: the client sends one JSON document and shuts down its write side;
the server answers with one JSON document and closes.

    {"Ping": 1}                                        -> {"Pong": 1}
    {"Kill": 1}                                        -> (server exits, no answer)
    {"Run": {"parameter": [...], "initial_guess": [...], "initial_lagrange_multipliers": [...],
             "initial_penalty": c}}                    (the last three optional)
      -> {"exit_status": "Converged", "num_outer_iterations": .., "num_inner_iterations": ..,
          "last_problem_norm_fpr": .., "delta_y_norm_over_c": .., "f2_norm": .., "solve_time_ms": ..,
          "penalty": .., "solution": [...], "lagrange_multipliers": [...], "cost": ..}
      -> {"type": "Error", "code": 1000|1600|1700|1800|2000, "message": "..."}  on failure

Extension (not in OpEn): ``{"RunBatch": {"parameter": [[...], ...]}}`` solves all rows in one launch and answers
``{"solutions": [<Run answer>, ...]}`` -- the reason for a GPU behind the socket.
"""
from __future__ import annotations

import json
import socket
import threading
import time
from typing import Callable, Optional

ERR_INVALID_REQUEST = 1000
ERR_WRONG_PARAMETER = 1600
ERR_WRONG_INITIAL_GUESS = 1700
ERR_WRONG_MULTIPLIERS = 1800
ERR_SOLVER = 2000


def _error(code: int, message: str) -> dict:
    return {"type": "Error", "code": code, "message": message}


CLIENT_TIMEOUT_S = 10.0   # per-connection receive / send timeout of the server


def _answer(sol) -> dict:
    return {"exit_status": sol.exit_status, "num_outer_iterations": sol.num_outer_iterations,
            "num_inner_iterations": sol.num_inner_iterations, "last_problem_norm_fpr": sol.last_problem_norm_fpr,
            "delta_y_norm_over_c": sol.f1_infeasibility, "f2_norm": sol.f2_norm, "solve_time_ms": sol.solve_time_ms,
            "penalty": sol.penalty, "solution": list(sol.solution),
            "lagrange_multipliers": list(sol.lagrange_multipliers), "cost": sol.cost}


def _is_number_list(x, length=None) -> bool:
    return isinstance(x, (list, tuple)) and (length is None or len(x) == length) and \
        all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in x)


def handle_request(solver, text: str):
    """One request document -> (answer dict or None, keep_running). Never raises: a malformed but valid-JSON request
    (wrong types, non-numeric entries) is answered with the 1000 "Invalid request" error like any other bad document."""
    try:
        return _handle_request(solver, text)
    except Exception as exc:   # the accept loop must survive whatever a client sends
        return _error(ERR_INVALID_REQUEST, f"Invalid request ({type(exc).__name__})"), True


def _handle_request(solver, text: str):
    try:
        req = json.loads(text)
    except ValueError:
        return _error(ERR_INVALID_REQUEST, "Invalid request"), True
    if not isinstance(req, dict) or len(req) != 1:
        return _error(ERR_INVALID_REQUEST, "Invalid request"), True
    if "Ping" in req:
        return {"Pong": 1}, True
    if "Kill" in req:
        return None, False
    n, np_ = solver.num_decision_variables, solver.num_parameters
    if "Run" in req and isinstance(req["Run"], dict) and "parameter" in req["Run"]:
        r = req["Run"]
        if not _is_number_list(r["parameter"]):
            return _error(ERR_INVALID_REQUEST, "Invalid request"), True
        for opt in ("initial_guess", "initial_lagrange_multipliers"):
            if r.get(opt) is not None and not _is_number_list(r[opt]):
                return _error(ERR_INVALID_REQUEST, "Invalid request"), True
        if r.get("initial_penalty") is not None and not isinstance(r["initial_penalty"], (int, float)):
            return _error(ERR_INVALID_REQUEST, "Invalid request"), True
        if len(r["parameter"]) != np_:
            return _error(ERR_WRONG_PARAMETER, f"wrong number of parameters: provided {len(r['parameter'])}, expected {np_}"), True
        u0, y0 = r.get("initial_guess"), r.get("initial_lagrange_multipliers")
        if u0 is not None and len(u0) != n:
            return _error(ERR_WRONG_INITIAL_GUESS, f"initial guess has incompatible dimensions: provided {len(u0)}, expected {n}"), True
        if y0 is not None and len(y0) != n:
            return _error(ERR_WRONG_MULTIPLIERS, f"wrong dimension of Langrange multipliers: provided {len(y0)}, expected {n}"), True
        sol = solver.run(r["parameter"], initial_guess=u0, initial_lagrange_multipliers=y0,
                         initial_penalty=r.get("initial_penalty"))
        if sol is None:
            return _error(ERR_SOLVER, "problem solution failed"), True
        return _answer(sol), True
    if "RunBatch" in req and isinstance(req["RunBatch"], dict) and "parameter" in req["RunBatch"]:
        rows = req["RunBatch"]["parameter"]
        if not isinstance(rows, list) or not all(_is_number_list(row) for row in rows):
            return _error(ERR_INVALID_REQUEST, "Invalid request"), True
        if any(len(row) != np_ for row in rows):
            return _error(ERR_WRONG_PARAMETER, f"wrong number of parameters: expected {np_} per row"), True
        sols = solver.run_many(rows)
        return {"solutions": [_answer(s) for s in sols]}, True
    return _error(ERR_INVALID_REQUEST, "Invalid request"), True


def serve(solver, ip: str = "127.0.0.1", port: int = 8333, ready: Optional[threading.Event] = None,
          bound: Optional[list] = None) -> None:
    """Blocking accept loop; returns after a ``Kill`` request. ``port=0`` picks a free port (reported via ``bound``)."""
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((ip, port))
    srv.listen(16)
    if bound is not None:
        bound.append(srv.getsockname()[1])
    if ready is not None:
        ready.set()
    running = True
    try:
        while running:
            conn, _ = srv.accept()
            with conn:
                conn.settimeout(CLIENT_TIMEOUT_S)     # a stalled client must not block the (single-threaded) server
                chunks = []
                try:
                    while True:
                        data = conn.recv(65536)
                        if not data:
                            break
                        chunks.append(data)
                    answer, running = handle_request(solver, b"".join(chunks).decode("utf-8", "replace"))
                    if answer is not None:
                        conn.sendall(json.dumps(answer).encode())
                except (socket.timeout, OSError):
                    continue                           # drop this connection, keep serving
    finally:
        srv.close()


# ------------------------------------------------------------------------------------------------------------
# Client side: the part of opengen.tcp the reference's tracker touches (start / ping / call / kill, is_ok / get).
class SolverStatus:
    def __init__(self, d: dict):
        self.exit_status = d["exit_status"]
        self.num_outer_iterations = d["num_outer_iterations"]
        self.num_inner_iterations = d["num_inner_iterations"]
        self.last_problem_norm_fpr = d["last_problem_norm_fpr"]
        self.f1_infeasibility = d["delta_y_norm_over_c"]
        self.f2_norm = d["f2_norm"]
        self.solve_time_ms = d["solve_time_ms"]
        self.penalty = d["penalty"]
        self.solution = d["solution"]
        self.cost = d["cost"]
        self.lagrange_multipliers = d["lagrange_multipliers"]


class SolverError:
    def __init__(self, d: dict):
        self.code, self.message = d["code"], d["message"]


class SolverResponse:
    def __init__(self, d: dict):
        self._r = SolverError(d) if d.get("type") == "Error" else SolverStatus(d)

    def is_ok(self) -> bool:
        return isinstance(self._r, SolverStatus)

    def get(self):
        return self._r

    def __getitem__(self, key):
        return getattr(self._r, key)


class OptimizerTcpManager:
    """``OptimizerTcpManager(optimizer_path)`` of opengen starts the generated server binary; here ``start()`` runs
    the accept loop on a daemon thread of this process around ``solver_factory()`` (the GPU handle lives in the
    calling process). With ``ip``/``port`` and no factory it attaches to a server that is already running
    (``python -m dyobav_mpcnwta_warehouse_amd.tcp config.yaml --port P``), as opengen does for a remote server."""

    def __init__(self, optimizer_path: Optional[str] = None, ip: Optional[str] = None, port: Optional[int] = None,
                 solver_factory: Optional[Callable] = None):
        self.optimizer_path = optimizer_path
        self.ip = ip or "127.0.0.1"
        self.port = port
        self._factory = solver_factory
        self._thread = None

    def start(self):
        if self._factory is None:
            if self.port is None:
                raise ValueError("no solver factory and no port of a running server")
            self.ping()
            return
        ready, bound = threading.Event(), []
        solver = self._factory()
        self._thread = threading.Thread(target=serve, args=(solver, self.ip, self.port or 0, ready, bound), daemon=True)
        self._thread.start()
        if not ready.wait(10.0):
            raise RuntimeError("TCP server did not start")
        self.port = bound[0]

    def _exchange(self, text: str, buffer_len: int = 4096, max_data_size: int = 1 << 26) -> str:
        with socket.create_connection((self.ip, self.port), timeout=60.0) as s:
            s.sendall(text.encode())
            s.shutdown(socket.SHUT_WR)
            data = b""
            while len(data) < max_data_size:
                chunk = s.recv(buffer_len)
                if not chunk:
                    break
                data += chunk
        return data.decode()

    def ping(self):
        return json.loads(self._exchange('{"Ping":1}'))

    def kill(self):
        try:
            self._exchange('{"Kill":1}')
        except OSError:
            pass
        if self._thread is not None:
            self._thread.join(5.0)
            self._thread = None

    def call(self, p, initial_guess=None, initial_y=None, initial_penalty=None, buffer_len=4096,
             max_data_size=1 << 26) -> SolverResponse:
        run = {"parameter": [float(v) for v in p]}
        if initial_guess is not None:
            run["initial_guess"] = [float(v) for v in initial_guess]
        if initial_y is not None:
            run["initial_lagrange_multipliers"] = [float(v) for v in initial_y]
        if initial_penalty is not None:
            run["initial_penalty"] = float(initial_penalty)
        return SolverResponse(json.loads(self._exchange(json.dumps({"Run": run}), buffer_len, max_data_size)))

    def call_batch(self, P) -> list:
        d = json.loads(self._exchange(json.dumps({"RunBatch": {"parameter": [[float(v) for v in row] for row in P]}})))
        if d.get("type") == "Error":
            return [SolverResponse(d)]
        return [SolverResponse(x) for x in d["solutions"]]


def main(argv=None):
    import argparse
    from .configs import CircularRobotSpecification, MpcConfiguration
    from .solver import make_config, solver
    ap = argparse.ArgumentParser(description="NMPC solver behind OpEn's TCP/JSON protocol")
    ap.add_argument("yaml")
    ap.add_argument("--ip", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8333)
    ap.add_argument("--dtype", default="float64")
    a = ap.parse_args(argv)
    s = solver(make_config(MpcConfiguration.from_yaml(a.yaml), CircularRobotSpecification.from_yaml(a.yaml)), dtype=a.dtype)
    print(f"[nmpc-tcp] serving on {a.ip}:{a.port}", flush=True)
    t0 = time.time()
    serve(s, a.ip, a.port)
    print(f"[nmpc-tcp] killed after {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
