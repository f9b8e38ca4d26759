"""Synthetic parameter batches for the batched NMPC solver.

Every instance is one flat parameter vector ``p`` in exactly the layout the reference's tracker assembles
(``/root/reference/src/pkg_mpc_tracker/trajectory_tracker.py:315-317``; symbol order
``solver_build/mpc_builder.py:47-60``)::

    u_m1(2) s_0(3) s_N(3) q(10) r_s(3N) r_v(N) c_0(3*Nother) c(3*N*Nother) o_s(12*Nstc) o_d(6*(N+1)*Ndyn)
    q_stc(N) q_dyn(N)

The generator is deterministic (``numpy.random.default_rng(seed)``) and vectorised, so the same batch can be
regenerated on the GPU box and in the authoring container.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class ParamLayout:
    """Offsets of the 12 parameter blocks."""
    N: int = 20
    Nother: int = 10
    Nstc: int = 10
    Ndyn: int = 15

    @property
    def um1(self):
        return 0

    @property
    def s0(self):
        return 2

    @property
    def sN(self):
        return 5

    @property
    def q(self):
        return 8

    @property
    def rs(self):
        return 18

    @property
    def rv(self):
        return self.rs + 3 * self.N

    @property
    def c0(self):
        return self.rv + self.N

    @property
    def c(self):
        return self.c0 + 3 * self.Nother

    @property
    def os(self):
        return self.c + 3 * self.N * self.Nother

    @property
    def od(self):
        return self.os + 12 * self.Nstc

    @property
    def qstc(self):
        return self.od + 6 * (self.N + 1) * self.Ndyn

    @property
    def qdyn(self):
        return self.qstc + self.N

    @property
    def np_(self):
        return self.qdyn + self.N


# work-mode tuning parameters of the shipped yaml files (trajectory_tracker.py:138-139 + config/mpc_*.yaml:33-43)
WORK_MODE_Q = (0.0, 10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 100.0, 10.0, 20.0)


def box_halfspaces(cx, cy, hx, hy):
    """(b, a0, a1) rows of an axis-aligned box, in the convention of
    ``pkg_mpc_tracker/utils_geo.py:35-62`` (A (p - centre) <= 1  <=>  b - a0 x - a1 y >= 0)."""
    a0 = np.stack([1.0 / hx, -1.0 / hx, np.zeros_like(hx), np.zeros_like(hx)], axis=-1)
    a1 = np.stack([np.zeros_like(hy), np.zeros_like(hy), 1.0 / hy, -1.0 / hy], axis=-1)
    b = a0 * cx[..., None] + a1 * cy[..., None] + 1.0
    return b, a0, a1


def make_batch(B: int, layout: ParamLayout = ParamLayout(), seed: int = 0, n_ped: int = 2, n_hyp: int = 5,
               ts: float = 0.2, base_speed: float = 1.2, n_boxes: int = 4, ped_mode: str = "toward_robot",
               dtype=np.float64) -> np.ndarray:
    """SURVEY.md 8d synthetic generator. Returns ``P[B, np]``.

    Robot: s0 xy ~ U(-10,10)^2, theta ~ U(-pi,pi); previous action u_m1 = (U(0,1.2), U(-0.3,0.3)).
    Reference: straight polyline from s0 along heading theta + U(-0.5,0.5), one point every ts*base_speed
    (what ``TrajectoryTracker.get_ref_traj`` produces in 'work' mode); r_v = base_speed.
    Pedestrians: start 3-8 m ahead, +-2 m lateral, walk toward the robot at 1.0-1.5 m/s; hypothesis h fans
    out by (h - (n_hyp-1)/2) * 0.15 rad; ellipse radii 0.2 + 0.05 t, angle 0, alpha 1
    (``main_base.py:293-302``); remaining Ndyn slots zero (``interfaces/mpc_interface.py:82-88``).
    ``ped_mode="oncoming"`` (not a BASELINE configuration; used by the parity tests) makes the pedestrians walk
    against the robot's heading +-0.3 rad instead of straight at it, which keeps the hard ellipse constraint
    feasible for more instances; ``ped_mode="passing"`` puts them on a parallel lane 1.5-4 m (x fan width) to the side (hard
    constraint feasible: the family where the solver converges, reported next to the contract family by bench.py).
    Static: ``n_boxes`` axis-aligned 1 x 2 m boxes within 6 m of the robot; remaining slots zero.
    Other robots: zero (reference default, ``trajectory_tracker.py:295-296``).
    """
    L = layout
    N = L.N
    assert n_ped * n_hyp <= L.Ndyn and n_boxes <= L.Nstc
    rng = np.random.default_rng(seed)
    P = np.zeros((B, L.np_), dtype=np.float64)

    xy = rng.uniform(-10.0, 10.0, size=(B, 2))
    th = rng.uniform(-np.pi, np.pi, size=B)
    P[:, L.um1] = rng.uniform(0.0, 1.2, size=B)
    P[:, L.um1 + 1] = rng.uniform(-0.3, 0.3, size=B)
    P[:, L.s0:L.s0 + 2] = xy
    P[:, L.s0 + 2] = th
    P[:, L.q:L.q + 10] = np.asarray(WORK_MODE_Q)

    # reference states: N rows (x, y, heading)
    hd = th + rng.uniform(-0.5, 0.5, size=B)
    d = np.stack([np.cos(hd), np.sin(hd)], axis=1)
    steps = (np.arange(1, N + 1) * ts * base_speed)[None, :, None]
    ref_xy = xy[:, None, :] + steps * d[:, None, :]
    rs = np.concatenate([ref_xy, np.broadcast_to(hd[:, None, None], (B, N, 1))], axis=2)
    P[:, L.rs:L.rs + 3 * N] = rs.reshape(B, 3 * N)
    P[:, L.sN:L.sN + 3] = rs[:, -1, :]
    P[:, L.rv:L.rv + N] = base_speed

    # static boxes -> half-space rows (b0..3, a0_0..3, a1_0..3)
    if n_boxes:
        ang = rng.uniform(-np.pi, np.pi, size=(B, n_boxes))
        rad = rng.uniform(1.5, 6.0, size=(B, n_boxes))
        bcx = xy[:, 0:1] + rad * np.cos(ang)
        bcy = xy[:, 1:2] + rad * np.sin(ang)
        tall = rng.random(size=(B, n_boxes)) < 0.5
        hx = np.where(tall, 0.5, 1.0)
        hy = np.where(tall, 1.0, 0.5)
        b, a0, a1 = box_halfspaces(bcx, bcy, hx, hy)
        os_ = np.concatenate([b, a0, a1], axis=-1)  # [B, n_boxes, 12]
        P[:, L.os:L.os + 12 * n_boxes] = os_.reshape(B, 12 * n_boxes)

    # pedestrians x hypotheses -> ellipses [Ndyn][N+1][6]
    od = np.zeros((B, L.Ndyn, N + 1, 6))
    if n_ped:
        fwd = np.stack([np.cos(th), np.sin(th)], axis=1)
        lat = np.stack([-np.sin(th), np.cos(th)], axis=1)
        ahead = rng.uniform(3.0, 8.0, size=(B, n_ped))
        side = rng.uniform(-2.0, 2.0, size=(B, n_ped))
        start = xy[:, None, :] + ahead[..., None] * fwd[:, None, :] + side[..., None] * lat[:, None, :]
        speed = rng.uniform(1.0, 1.5, size=(B, n_ped))
        if ped_mode == "toward_robot":
            to_robot = xy[:, None, :] - start
            base_ang = np.arctan2(to_robot[..., 1], to_robot[..., 0])
        elif ped_mode == "oncoming":
            base_ang = (th[:, None] + np.pi) + rng.uniform(-0.3, 0.3, size=(B, n_ped))
        elif ped_mode == "passing":
            # pedestrians pass on a parallel lane 1.5-4 m (x fan width) to the side, against the robot's heading: the obstacle terms
            # are exercised (soft margins are touched now and then) while the hard constraint stays feasible -- the
            # operating point of the reference's warehouse runs, where the solver converges
            # (offset scaled with the width of the hypothesis fan, (n_hyp - 1) * 0.075 rad to either side)
            lane_off = rng.uniform(1.5, 4.0, size=(B, n_ped)) * max(1.0, (n_hyp - 1) / 4.0) \
                * np.where(rng.random(size=(B, n_ped)) < 0.5, -1.0, 1.0)
            start = xy[:, None, :] + ahead[..., None] * fwd[:, None, :] + lane_off[..., None] * lat[:, None, :]
            base_ang = np.broadcast_to(th[:, None] + np.pi, (B, n_ped)).copy()
        else:
            raise ValueError(f"unknown ped_mode {ped_mode!r}")
        t = np.arange(N + 1)[None, None, None, :]
        fan = (np.arange(n_hyp) - (n_hyp - 1) / 2.0) * 0.15
        a = base_ang[:, :, None] + fan[None, None, :]           # [B, ped, hyp]
        dist = speed[:, :, None, None] * t * ts                  # [B, ped, 1, T]
        cxs = start[:, :, None, None, 0] + dist * np.cos(a)[..., None]
        cys = start[:, :, None, None, 1] + dist * np.sin(a)[..., None]
        r = 0.2 + 0.05 * np.arange(N + 1)
        k = n_ped * n_hyp
        od[:, :k, :, 0] = cxs.reshape(B, k, N + 1)
        od[:, :k, :, 1] = cys.reshape(B, k, N + 1)
        od[:, :k, :, 2] = r
        od[:, :k, :, 3] = r
        od[:, :k, :, 4] = 0.0
        od[:, :k, :, 5] = 1.0
    P[:, L.od:L.od + 6 * (N + 1) * L.Ndyn] = od.reshape(B, -1)

    P[:, L.qstc:L.qstc + N] = 10.0
    P[:, L.qdyn:L.qdyn + N] = 10.0
    return P.astype(dtype)


# BASELINE.json configs[1..4] -> (layout, generator kwargs, batch)
BENCH_CONFIGS = {
    # batch=1024 random init states, N=20, 2 obstacles x 5 WTA hypotheses, fp32
    "cfg1_b1024_n20_2x5": dict(layout=ParamLayout(20, 10, 10, 15), B=1024, seed=0, n_ped=2, n_hyp=5),
    # batch=65536 main_eva scenarios, N=20, 4 obs x 10 hypotheses
    "cfg2_b65536_n20_4x10": dict(layout=ParamLayout(20, 10, 10, 40), B=65536, seed=1, n_ped=4, n_hyp=10),
    # long horizon N=40, 8 obs x 20 hypotheses, batch=8192
    "cfg4_b8192_n40_8x20": dict(layout=ParamLayout(40, 10, 10, 160), B=8192, seed=5, n_ped=8, n_hyp=20),
}


def make_batch_chunked(B: int, layout: ParamLayout = ParamLayout(), seed: int = 0, chunk: int = 8192, dtype=np.float32,
                       **kw) -> np.ndarray:
    """``make_batch`` for large B with bounded host memory: the batch is generated ``chunk`` instances at a time (chunk
    i from the seed sequence ``(seed, i)``) straight into the output array. Deterministic in (seed, B, chunk); for
    B <= chunk it is ``make_batch(B, ..., seed=seed)`` itself. Used by bench.py, where eight ranks generate 65 536
    instances each on one host (the float64 intermediates of one call would be ~8 GB per rank)."""
    if B <= chunk:
        return make_batch(B, layout, seed=seed, dtype=dtype, **kw)
    out = np.empty((B, layout.np_), dtype=dtype)
    for i, lo in enumerate(range(0, B, chunk)):
        hi = min(B, lo + chunk)
        sub = int(np.random.SeedSequence([seed, i]).generate_state(1)[0])
        out[lo:hi] = make_batch(hi - lo, layout, seed=sub, dtype=dtype, **kw)
    return out


def make_config_batch(name: str, B: int | None = None, seed: int | None = None, dtype=np.float64):
    cfg = dict(BENCH_CONFIGS[name])
    if B is not None:
        cfg["B"] = B
    if seed is not None:
        cfg["seed"] = seed
    layout = cfg.pop("layout")
    return layout, make_batch(layout=layout, dtype=dtype, **cfg)


# ---------------------------------------------------------------------------------------------------------------------
# Closed-loop scenarios: the inputs of evaluate.BatchEvaluator (row f3) for B Monte-Carlo warehouse runs, and the
# parameter batches harvested from them -- the distribution the reference's own evaluation loop produces
# (main_eva.py:6-14 -> MainBase.run, main_base.py:448-464: max_num_run repetitions of a scenario with staggering
# pedestrians; main_base.py:293-302: the obstacle tensor of every time step), as opposed to make_batch's one-shot draw.
# ---------------------------------------------------------------------------------------------------------------------
def make_closed_loop_scenarios(B: int, seed: int = 13, n_ped: int = 4, n_boxes: int = 14) -> dict:
    """B corridor scenarios in the style of the reference's ``scenario_0..2`` (main_base.py:38-57: a robot driving an
    aisle, pedestrians crossing it on way-point paths): the robot starts near the origin heading along +x towards a goal
    ~8 m ahead, ``n_boxes`` shelf blocks line the aisle on both sides, ``n_ped`` pedestrians start 5-10 m ahead on
    either side and walk two way-points across / against the aisle (HUMAN_VMAX with the seeded stagger of
    basic_agent.py:64-82 is applied by the evaluator). Deterministic in (B, seed, n_ped, n_boxes); returns the keyword
    arguments of ``BatchEvaluator`` as numpy arrays / lists."""
    rng = np.random.default_rng(seed)
    boxes = []
    for i in range(n_boxes):
        c = np.array([1.5 + 1.1 * i, (-1) ** i * rng.uniform(1.6, 2.6)])
        hx, hy = rng.uniform(0.3, 0.6, 2)
        boxes.append([[c[0] + hx, c[1] + hy], [c[0] - hx, c[1] + hy], [c[0] - hx, c[1] - hy], [c[0] + hx, c[1] - hy]])
    starts = np.stack([np.zeros(B), rng.uniform(-0.4, 0.4, B), rng.uniform(-0.3, 0.3, B)], axis=1)
    goals = np.stack([8.0 + rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)], axis=1)
    paths = [[(float(g[0]), float(g[1]))] for g in goals]
    hstart = np.empty((B, n_ped, 2))
    hpath = np.empty((B, n_ped, 2, 2))
    for h in range(n_ped):
        side = 1.0 if h % 2 == 0 else -1.0
        x0 = rng.uniform(5.0, 10.0, B)
        y0 = side * rng.uniform(2.0, 3.5, B)
        hstart[:, h] = np.stack([x0, y0], axis=1)
        # first way-point across the aisle and towards the robot, second further down the aisle on the other side
        w1 = hstart[:, h] + np.stack([rng.uniform(-3.5, -2.0, B), -side * rng.uniform(4.5, 5.5, B)], axis=1)
        w2 = w1 + np.stack([rng.uniform(-3.5, -2.0, B), -side * rng.uniform(0.0, 1.0, B)], axis=1)
        hpath[:, h, 0], hpath[:, h, 1] = w1, w2
    return dict(robot_starts=starts, robot_paths=paths, human_starts=hstart, human_paths=hpath,
                map_polygons=np.array(boxes))


_WAREHOUSE = None


def warehouse_world() -> dict:
    """The warehouse the reference evaluates in, as recorded from the reference (``data/warehouse_world.json``, written by
    ``tests/golden/make_golden.py``: node graph of ``data/warehouse_sim_original/mygraph.json``, the 55 inflated static
    rectangles its map pipeline extracts from ``mymap.pgm`` (main_base.py:123-127), and ``scenario_0..2`` of
    main_base.py:36-58 -- all in world coordinates through the reference's own transform)."""
    global _WAREHOUSE
    if _WAREHOUSE is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "warehouse_world.json")) as fh:
            _WAREHOUSE = json.load(fh)
    return _WAREHOUSE


def make_reference_scenarios(B: int, seed: int = 13, n_ped: int = 4, n_waypoints: int = 4, scenario: int | None = None) -> dict:
    """B Monte-Carlo runs of the REFERENCE's evaluation scenarios (main_eva.py:6-14 -> MainBase.run, main_base.py:448-464:
    ``max_num_run`` repetitions of ``scenario(SCENARIO_NUM)`` with staggering pedestrians) on the reference's warehouse:

    * run b is ``scenario_{b % 3}`` (or ``scenario`` for all): the robot's start state and node path exactly as
      main_base.py:36-58 / ``_prepare_agents`` (:129-150) give them, the 55 static rectangles of the warehouse map;
    * pedestrian 0 is the scenario's own pedestrian (its start point and node path);
    * pedestrians 1 .. n_ped - 1 -- BASELINE configs[2] has four -- walk other node paths of the same graph: a seeded
      non-backtracking walk of ``n_waypoints`` nodes that starts at most two edges away from a node of the robot's path
      (and at least 3 m from the robot's start), from a point within 0.5 m of its first node;
    * every pedestrian moves at HUMAN_VMAX with the stagger of basic_agent.py:64-82 (``human_stagger`` =
      HUMAN_STAGGER = 0.5 is the evaluator's argument; seeded per run).

    Deterministic in (B, seed, n_ped, n_waypoints, scenario). Returns the keyword arguments of ``BatchEvaluator`` plus
    ``scenario_index`` [B]."""
    w = warehouse_world()
    nodes = {int(k): np.array(v, dtype=float) for k, v in w["nodes_world"].items()}
    adj: dict = {k: [] for k in nodes}
    for a, b in w["graph_edges"]:
        adj[a].append(b)
        adj[b].append(a)
    rng = np.random.default_rng(seed)
    W = n_waypoints
    near = {}          # scenario -> candidate first nodes of the extra pedestrians
    for k, sc in w["scenarios"].items():
        reach = set(sc["robot_path"])
        for _ in range(2):
            reach |= {n for r in list(reach) for n in adj[r]}
        st = np.array(sc["robot_start_world"][:2])
        near[int(k)] = sorted(n for n in reach if np.linalg.norm(nodes[n] - st) >= 3.0)
    sidx = np.full(B, scenario, dtype=np.int64) if scenario is not None else np.arange(B) % 3
    starts = np.empty((B, 3))
    paths = []
    hstart = np.empty((B, n_ped, 2))
    hpath = np.empty((B, n_ped, W, 2))
    for b in range(B):
        sc = w["scenarios"][str(int(sidx[b]))]
        starts[b] = sc["robot_start_world"]
        paths.append([tuple(p) for p in sc["robot_path_world"]])
        own = [np.array(p) for p in sc["human_paths_world"][0]]
        hstart[b, 0] = sc["human_starts_world"][0]
        hpath[b, 0] = np.stack((own + [own[-1]] * W)[:W])
        for h in range(1, n_ped):
            cand = near[int(sidx[b])]
            walk = [cand[int(rng.integers(len(cand)))]]
            while len(walk) < W + 1:
                nxt = [n for n in adj[walk[-1]] if len(walk) < 2 or n != walk[-2]] or adj[walk[-1]]
                walk.append(nxt[int(rng.integers(len(nxt)))])
            hstart[b, h] = nodes[walk[0]] + rng.uniform(-0.5, 0.5, 2)
            hpath[b, h] = np.stack([nodes[n] for n in walk[1:]])
    return dict(robot_starts=starts, robot_paths=paths, human_starts=hstart, human_paths=hpath,
                map_polygons=np.array(w["map_polygons_world"], dtype=float), scenario_index=sidx)


HUMAN_STAGGER = 0.5     # main_base.py:77


def harvest_closed_loop(config, B: int, steps=(1, 8, 20), seed: int = 13, n_ped: int = 4, n_hyp: int = 10,
                        dtype=np.float32, human_stagger: float | None = None, return_device: bool = False,
                        family: str = "corridor"):
    """Parameter vectors ``P[B, np]`` as the closed loop produces them: ``make_closed_loop_scenarios(B, seed, n_ped)``
    advanced by ``evaluate.BatchEvaluator`` (row f3, pinned to the reference) with ``n_hyp`` hypotheses per pedestrian
    fanned around the constant-velocity prediction (``config.Ndynobs`` >= n_ped * n_hyp), and the assembled
    parameter vector of scenario b captured at time step ``steps[b % len(steps)]`` (reference family: ``steps[(b // 3) %
    len(steps)]``: every scenario at every step) --
    or at the last earlier capture step it was still running. ``family``: ``"reference"`` = ``make_reference_scenarios``
    (scenario_0..2 on the warehouse map, HUMAN_STAGGER 0.5), ``"corridor"`` = ``make_closed_loop_scenarios`` (round 5).
    Needs the GPU (the closed loop solves on the device).
    Returns ``(P, step_of_row)``: numpy arrays, or torch device tensors with ``return_device``."""
    import copy

    import torch

    from .evaluate import BatchEvaluator
    steps = tuple(sorted(int(s) for s in steps))
    if family == "reference":      # the reference's own scenarios on its warehouse map, HUMAN_STAGGER as main_base.py:77
        sc = make_reference_scenarios(B, seed=seed, n_ped=n_ped)
        sc.pop("scenario_index")
        human_stagger = HUMAN_STAGGER if human_stagger is None else human_stagger
    elif family == "corridor":     # the builder-designed corridor family of round 5
        sc = make_closed_loop_scenarios(B, seed=seed, n_ped=n_ped)
        human_stagger = 0.2 if human_stagger is None else human_stagger
    else:
        raise ValueError(f"family = {family!r} (reference or corridor)")
    cfg = copy.copy(config)
    cfg.max_active_dynobs = n_ped * n_hyp
    ev = BatchEvaluator(cfg, dtype=dtype, human_stagger=human_stagger, seed=seed, n_hyp=n_hyp, **sc)
    ev.time_solves = False
    out = torch.zeros(B, ev.h.np_, dtype=ev.tdt, device=ev.dev)
    step_of = torch.full((B,), -1, dtype=torch.int32, device=ev.dev)
    ns = len(steps)
    # capture slot of scenario b: b % ns -- for the reference family (b // 3) % ns, so that every one of its three
    # scenarios (b % 3) is captured at every step
    slot = torch.arange(B, device=ev.dev)
    slot = (slot // 3) % ns if family == "reference" else slot % ns

    def grab(kt, idx, Pa):
        if kt not in steps:
            return
        i = steps.index(kt)
        rows = idx if idx is not None else torch.arange(B, device=ev.dev)
        # slot i is the capture step of the scenarios with slot[b] == i; a scenario whose own step comes later takes every
        # earlier capture step too and is overwritten until its own -- so one that stops running in between keeps the LAST
        # capture step it was still running at (`| step_of < 0` kept the first)
        take = slot[rows] >= i
        rsel = rows[take]
        out[rsel] = Pa[:rows.numel()][take]
        step_of[rsel] = kt

    ev.on_params = grab
    ev.run(max_steps=steps[-1] + 1)
    ev.close()
    if steps[0] == 0 or bool((step_of >= 0).all()):
        pass
    else:       # (a scenario that is over before the first capture step would leave an all-zero parameter row behind)
        raise RuntimeError(f"{int((step_of < 0).sum())} scenarios ended before the first capture step {steps[0]}")
    if return_device:
        return out, step_of
    return out.cpu().numpy(), step_of.cpu().numpy()
