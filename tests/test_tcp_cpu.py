"""CPU tests of the TCP/JSON front end (row f4): wire format, error codes and the tracker's use_tcp path, with a
scripted solver behind the socket (no GPU). The protocol is restated from opengen's published client/server pair and
is not pinned by anything in the reference ("wire format unpinned", see the module docstring)."""
import json
import os
import socket
import types

import numpy as np
import pytest

from dyobav_mpcnwta_warehouse_amd import tcp
from dyobav_mpcnwta_warehouse_amd.configs import CircularRobotSpecification, MpcConfiguration
from dyobav_mpcnwta_warehouse_amd.motion_model import UnicycleModel
from dyobav_mpcnwta_warehouse_amd.solver import OptimizerSolution, shift_solution
from dyobav_mpcnwta_warehouse_amd.trajectory_tracker import TrajectoryTracker

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config", "mpc_fast.yaml")


class _Scripted:
    num_parameters, num_decision_variables = 2778, 40

    def __init__(self):
        self.calls = []

    def _sol(self, p, n):
        u = [0.5 + 0.02 * k + 0.1 * n if j == 0 else 0.1 - 0.01 * k for k in range(20) for j in range(2)]
        return OptimizerSolution(exit_status="Converged", num_outer_iterations=2, num_inner_iterations=7 + n,
                                 last_problem_norm_fpr=1e-5, f1_infeasibility=2e-5, f2_norm=0.0, solve_time_ms=1.25,
                                 penalty=50.0, solution=u, lagrange_multipliers=[0.5] * 40, cost=12.5 * n + float(p[0]))

    def run(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None):
        self.calls.append((initial_guess, initial_lagrange_multipliers, initial_penalty))
        if p[0] == -999.0:
            return None
        return self._sol(p, len(self.calls))

    def run_many(self, P):
        return [self._sol(p, i) for i, p in enumerate(P)]


@pytest.fixture
def manager():
    fake = _Scripted()
    mng = tcp.OptimizerTcpManager("mpc_solver/navi_fast", solver_factory=lambda: fake)
    mng.start()
    yield mng, fake
    mng.kill()


def _raw(mng, text):
    with socket.create_connection((mng.ip, mng.port), timeout=10) as s:
        s.sendall(text.encode())
        s.shutdown(socket.SHUT_WR)
        data = b""
        while True:
            c = s.recv(4096)
            if not c:
                break
            data += c
    return json.loads(data.decode())


def test_ping_run_and_optional_fields(manager):
    mng, fake = manager
    assert mng.ping() == {"Pong": 1}
    p = [0.25] + [0.0] * 2777
    r = mng.call(p)
    assert r.is_ok()
    s = r.get()
    assert s.exit_status == "Converged" and s.num_inner_iterations == 8 and s.cost == 12.5 + 0.25
    assert s.f1_infeasibility == 2e-5 and s.penalty == 50.0 and len(s.solution) == 40 and len(s.lagrange_multipliers) == 40
    assert r["solve_time_ms"] == 1.25
    assert fake.calls[-1] == (None, None, None)
    r = mng.call(p, initial_guess=[0.1] * 40, initial_y=[0.0] * 40, initial_penalty=25)
    assert r.is_ok() and fake.calls[-1] == ([0.1] * 40, [0.0] * 40, 25.0)
    # the raw document a non-Python caller sends / receives (field names of the generated server)
    d = _raw(mng, '{"Run" : {"parameter": [' + ",".join(map(str, p)) + ']}}')
    assert set(d) == {"exit_status", "num_outer_iterations", "num_inner_iterations", "last_problem_norm_fpr",
                      "delta_y_norm_over_c", "f2_norm", "solve_time_ms", "penalty", "solution", "lagrange_multipliers",
                      "cost"}


def test_error_documents(manager):
    mng, _ = manager
    r = mng.call([0.0] * 10)
    assert not r.is_ok() and r.get().code == 1600 and "parameters" in r.get().message
    p = [0.0] * 2778
    assert mng.call(p, initial_guess=[0.0] * 3).get().code == 1700
    assert mng.call(p, initial_y=[0.0] * 3).get().code == 1800
    assert mng.call([-999.0] + [0.0] * 2777).get().code == 2000
    for bad in ('{"Walk": 1}', "not json", '{"Run": 5}', "[1, 2]"):
        d = _raw(mng, bad)
        assert d["type"] == "Error" and d["code"] == 1000
    assert mng.ping() == {"Pong": 1}                     # the server survives bad requests


def test_malformed_but_valid_json_requests_do_not_kill_the_server(manager):
    """requests that parse as JSON but carry the wrong types used to raise inside the accept loop (TypeError
    in len(5), ValueError for non-numeric entries) and leave later clients hanging. They are answered with error 1000."""
    mng, _ = manager
    p = ",".join(["0.0"] * 2778)
    for bad in ('{"Run": {"parameter": 5}}', '{"Run": {"parameter": "abc"}}',
                '{"Run": {"parameter": [' + ",".join(['"x"'] * 2778) + ']}}',
                '{"Run": {"parameter": [' + p + '], "initial_guess": 7}}',
                '{"Run": {"parameter": [' + p + '], "initial_penalty": "big"}}',
                '{"RunBatch": {"parameter": 3}}', '{"RunBatch": {"parameter": [[1, 2], "row"]}}',
                '{"Run": {"parameter": [true, false]}}'):
        d = _raw(mng, bad)
        assert d["type"] == "Error" and d["code"] == 1000, bad[:40]
    assert mng.ping() == {"Pong": 1}
    assert mng.call([0.25] + [0.0] * 2777).is_ok()


def test_stalled_client_is_dropped(manager, monkeypatch):
    """A client that connects and never finishes its request is dropped after the receive timeout instead of blocking
    the single-threaded server for good."""
    mng, _ = manager
    monkeypatch.setattr(tcp, "CLIENT_TIMEOUT_S", 0.3)
    stalled = socket.create_connection((mng.ip, mng.port), timeout=10)
    stalled.sendall(b'{"Run": {"parameter": [0.0, ')          # ... and nothing more, no shutdown
    try:
        assert mng.ping() == {"Pong": 1}                      # served once the stalled connection has timed out
    finally:
        stalled.close()


def test_batch_request_extension(manager):
    mng, _ = manager
    P = [[float(i)] + [0.0] * 2777 for i in range(5)]
    out = mng.call_batch(P)
    assert len(out) == 5 and all(r.is_ok() for r in out)
    assert [r.get().cost for r in out] == [12.5 * i + i for i in range(5)]
    assert not mng.call_batch([[0.0] * 7])[0].is_ok()


def test_tracker_over_tcp_equals_tracker_in_process():
    """reference trajectory_tracker.py:385-400 vs :361-383: same numbers whichever way the solver is reached."""
    mpc, rob = MpcConfiguration.from_yaml(CFG), CircularRobotSpecification.from_yaml(CFG)
    outs = []
    for use_tcp in (False, True):
        tr = TrajectoryTracker(mpc, rob, use_tcp=use_tcp, verbose=False, solver_factory=_Scripted)
        tr.load_motion_model(UnicycleModel(rob.ts))
        tr.load_init_states(np.array([0.0, 0.0, 0.0]), np.array([5.0, 3.0, 1.57]))
        tr.set_work_mode("work")
        tr.set_ref_trajectory([(5.0, 0.0), (5.0, 3.0)])
        steps = [tr.run_step([0.0] * 120, [0.0] * 1890, mode="work") for _ in range(3)]
        outs.append((steps, np.array(tr.past_states), tr.cost_timelist))
        if use_tcp:
            tr.mng.kill()
    for (a, p, r, c), (a2, p2, r2, c2) in zip(outs[0][0], outs[1][0]):
        np.testing.assert_array_equal(np.array(a), np.array(a2))
        np.testing.assert_array_equal(np.array(p), np.array(p2))
        np.testing.assert_array_equal(r, r2)
        assert c == c2
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]


def test_tracker_over_tcp_raises_and_kills_on_error():
    mpc, rob = MpcConfiguration.from_yaml(CFG), CircularRobotSpecification.from_yaml(CFG)
    tr = TrajectoryTracker(mpc, rob, use_tcp=True, verbose=False, solver_factory=_Scripted)
    tr.load_motion_model(UnicycleModel(rob.ts))
    with pytest.raises(RuntimeError, match=r"\[1600\]"):
        tr.run_solver([0.0] * 5, np.zeros(3))
    with pytest.raises(OSError):
        tr.mng.ping()                                    # the server was told to stop (reference :396)


def test_shift_solution():
    U = np.arange(12.0).reshape(2, 6)
    np.testing.assert_array_equal(shift_solution(U), [[2, 3, 4, 5, 4, 5], [8, 9, 10, 11, 10, 11]])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is only in the authoring container")
def test_the_references_own_tcp_consumer_accepts_our_answers():
    """What CAN be pinned of row f4 without opengen: the consumer side. The REFERENCE's TrajectoryTracker.run_solver_tcp
    (pkg_mpc_tracker/trajectory_tracker.py:385-416, imported from /root/reference with stub `opengen` / `casadi` modules
    for its import lines) is run against this repo's OptimizerTcpManager + server -- `.call(parameters)`, `.is_ok()`,
    `.get().solution / .cost / .exit_status / .solve_time_ms`, and on an error `.get().code / .message` + `.kill()` --
    and must return exactly what its in-process twin run_solver returns for the same solver answers."""
    import importlib
    import sys
    saved = {k: sys.modules.get(k) for k in ("opengen", "opengen.opengen", "opengen.opengen.tcp", "opengen.opengen.tcp.solver_status",
                                             "casadi", "casadi.casadi", "configs", "pkg_mpc_tracker", "pkg_mpc_tracker.trajectory_tracker")}
    og = types.ModuleType("opengen")
    og.__path__ = []
    sub = types.ModuleType("opengen.opengen")
    sub.__path__ = []
    tcpm = types.ModuleType("opengen.opengen.tcp")
    tcpm.__path__ = []
    ss = types.ModuleType("opengen.opengen.tcp.solver_status")
    ss.SolverStatus = object
    og.opengen, sub.tcp, tcpm.solver_status = sub, tcpm, ss
    og.tcp = types.SimpleNamespace(OptimizerTcpManager=tcp.OptimizerTcpManager)
    cs = types.ModuleType("casadi")
    cs.__path__ = []
    csc = types.ModuleType("casadi.casadi")
    csc.SX = type("SX", (), {})
    cs.casadi = csc
    sys.modules.update({"opengen": og, "opengen.opengen": sub, "opengen.opengen.tcp": tcpm,
                        "opengen.opengen.tcp.solver_status": ss, "casadi": cs, "casadi.casadi": csc})
    sys.path.insert(0, "/root/reference/src")
    try:
        for k in ("configs", "pkg_mpc_tracker", "pkg_mpc_tracker.trajectory_tracker"):
            sys.modules.pop(k, None)
        ref_tt = importlib.import_module("pkg_mpc_tracker.trajectory_tracker")
        ref_mm = importlib.import_module("basic_motion_model.motion_model")
        fake = _Scripted()
        mng = tcp.OptimizerTcpManager("mpc_solver/navi_fast", solver_factory=lambda: fake)
        mng.start()
        me = types.SimpleNamespace(mng=mng, solver=_Scripted(), use_tcp=False, nu=2, ts=0.2,
                                   motion_model=ref_mm.UnicycleModel(0.2, rk4=True))
        p, state = [1.5] + [0.0] * 2777, np.array([1.0, 2.0, 0.3])
        over_tcp = ref_tt.TrajectoryTracker.run_solver_tcp(me, p, state)
        in_proc = ref_tt.TrajectoryTracker.run_solver(me, p, state)
        for a, b in zip(over_tcp[:3], in_proc[:3]):                  # taken_states, pred_states, actions
            np.testing.assert_array_equal(np.array(a), np.array(b))
        assert over_tcp[3] == in_proc[3] and over_tcp[5] == in_proc[5] == "Converged"      # cost, exit status
        assert over_tcp[4] == 1.25                                   # solve_time_ms as the server reported it
        with pytest.raises(RuntimeError, match=r"MPC Solver error: \[1600\]"):
            ref_tt.TrajectoryTracker.run_solver_tcp(me, [0.0] * 5, state)
        with pytest.raises(OSError):
            mng.ping()                                               # the reference killed the server (:396)
    finally:
        sys.path.remove("/root/reference/src")
        for k in ("configs", "pkg_mpc_tracker", "pkg_mpc_tracker.trajectory_tracker", "basic_motion_model", "basic_motion_model.motion_model"):
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ---- the RECALLED wire documents (tests/recalled/tcp_wire.json: not reference-backed, see tests/recalled/README.md) ----------------------------------------
def _wire():
    return json.load(open(os.path.join(ROOT, "tests", "recalled", "tcp_wire.json")))


def _run_bytes(w, p, u0=None, y0=None, c0=None) -> bytes:
    """the request exactly as opengen's client assembles it (string concatenation of map(str, .), NOT json.dumps)"""
    t = w["requests"]["run_template"]
    s = t["prefix"] + t["join"].join(map(str, p)) + "]"
    if u0 is not None:
        s += t["initial_guess"] + t["join"].join(map(str, u0)) + "]"
    if y0 is not None:
        s += t["initial_lagrange_multipliers"] + t["join"].join(map(str, y0)) + "]"
    if c0 is not None:
        s += t["initial_penalty"] + str(float(c0))
    return (s + "}}").encode()


def _raw_bytes(mng, payload: bytes) -> bytes:
    with socket.create_connection((mng.ip, mng.port), timeout=10) as s:
        s.sendall(payload)
        s.shutdown(socket.SHUT_WR)
        data = b""
        while True:
            c = s.recv(4096)
            if not c:
                break
            data += c
    return data


def test_recalled_wire_bytes_against_the_server(manager):
    """The exact request bytes opengen's client writes (recalled; each with its opengen source named in the fixture) over a
    raw socket, and the shape of what comes back: field names IN ORDER, JSON types, status strings, error documents."""
    mng, fake = manager
    w = _wire()
    assert json.loads(_raw_bytes(mng, w["requests"]["ping"]["bytes"].encode())) == w["responses"]["pong"]["document"]
    p = [0.25, 1e-05, -0.0] + [0.0] * 2775
    req = _run_bytes(w, p, u0=[0.1] * 40, y0=[0.0] * 40, c0=25)
    assert req.startswith(b'{"Run" : {"parameter": [0.25,1e-05,-0.0,0.0,') and req.endswith(b', "initial_penalty": 25.0}}')
    ans = _raw_bytes(mng, req)
    d = json.loads(ans, object_pairs_hook=list)                      # keeps the order of the fields on the wire
    assert [k for k, _ in d] == w["responses"]["solution_fields_in_order"]
    types = {"str": str, "int": int, "float": float}
    for k, v in d:
        t = w["responses"]["solution_field_types"][k]
        if t.startswith("list"):
            assert isinstance(v, list) and len(v) == 40 and all(isinstance(x, float) for x in v), k
        else:
            assert type(v) is types[t], (k, v)
    assert dict(d)["exit_status"] in w["responses"]["exit_status_values"]
    assert fake.calls[-1] == ([0.1] * 40, [0.0] * 40, 25.0)
    # error documents: fields in order, the template's codes and messages
    for code, payload in ((1600, _run_bytes(w, [0.0] * 7)), (1700, _run_bytes(w, p, u0=[0.0] * 3)),
                          (1800, _run_bytes(w, p, y0=[0.0] * 3)), (2000, _run_bytes(w, [-999.0] + [0.0] * 2777)),
                          (1000, b'{"Run" : {"parameter": [0.1,')):
        e = json.loads(_raw_bytes(mng, payload), object_pairs_hook=list)
        assert [k for k, _ in e] == w["responses"]["error_fields_in_order"]
        e = dict(e)
        assert e["type"] == "Error" and e["code"] == code and w["responses"]["errors"][str(code)] in e["message"], e
    # Kill: no answer, the server is gone afterwards
    assert _raw_bytes(mng, w["requests"]["kill"]["bytes"].encode()) == b""
