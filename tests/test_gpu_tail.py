"""Tail hand-off (nmpc_config.tail_latency, round 6 / VERDICT r5 item 4): once the last throughput launch of a solve is in its
drain phase -- every workgroup dispatched, at most `tail_latency` instances still running, each alone on its SIMD -- an
instance parks at its next outer-iteration boundary and the latency family's TAIL member (the speculative line search over
six wavefronts with the throughput kernels' own, gated evaluation) finishes it. The member returns the throughput kernels'
bits, so who solves which part of an instance is pure scheduling: every result array is identical with and without the
hand-off, for any threshold, behind the pilot's ranking and under a caller's dispatch order, on both members of the kernel
pairs, with an evaluation budget -- although WHICH instances are handed over depends on timing."""
import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm

pytestmark = pytest.mark.gpu

KEYS = ("U", "cost", "status", "iters")


def _cfg(lay, hint, **ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = hint
    cfg.latency_waves = 1                     # the throughput family (the automatic choice for these batch sizes anyway)
    for k, v in ov.items():
        setattr(cfg, k, v)
    return cfg


def _same(a, b, what):
    for k in KEYS:
        assert np.array_equal(a[k], b[k], equal_nan=True), (what, k, int((a[k] != b[k]).sum()))
    # info[:, :6]: residuals, penalty, evaluation counts -- results; [6], [7] are launch diagnostics (exchange rounds, W)
    assert np.array_equal(a["info"][:, :6], b["info"][:, :6], equal_nan=True), what


@pytest.mark.parametrize("dims,hint,slots", [((20, 10, 10, 40), 40, 14), ((20, 10, 10, 15), 10, 4), ((20, 10, 10, 15), 0, 6)],
                         ids=["cfg2-14slot", "cfg1-4slot", "cfg1-6slot"])
def test_tail_handoff_is_bit_identical(dims, hint, slots):
    lay = nm.scenarios.ParamLayout(*dims)
    n_ped, n_hyp = (4, 10) if dims[3] == 40 else (2, 5)
    B = 16384
    P = np.concatenate([nm.scenarios.make_batch_chunked(B // 2, lay, seed=41, n_ped=n_ped, n_hyp=n_hyp, ped_mode="passing", dtype=np.float32),
                        nm.scenarios.make_batch_chunked(B // 2, lay, seed=42, n_ped=n_ped, n_hyp=n_hyp, dtype=np.float32)])
    assert nm.layout_info(_cfg(lay, hint)).reg_slots_f32 == slots
    with nm.Handle(_cfg(lay, hint, tail_latency=-1, staged=1)) as h:
        ref = h.solve(P)
        assert h.last_launch_info()["tail_handed_off"] == 0 and h.last_launch_info()["staged_outer_iterations"] == 1
    # behind the pilot of the resumable solve (thresholds from a handful to "park almost from the start")
    handed_any = 0
    for thr in (48, 1024, 0):
        with nm.Handle(_cfg(lay, hint, tail_latency=thr, staged=1)) as h:
            r = h.solve(P)
            li = h.last_launch_info()
            assert li["tail_handed_off"] == (thr or li["tail_handed_off"]) and li["tail_handed_off"] > 0, li
            assert li["family"] == "throughput" and li["staged_outer_iterations"] == 1
        _same(r, ref, ("staged", thr))
        handed = r["info"][:, 7] > 0          # (info[7] = wavefronts per instance: > 0 where the latency family finished it)
        assert handed.sum() <= li["tail_handed_off"]
        handed_any += int(handed.sum())
    assert handed_any > 0                      # (the hand-off really happens: some instance was finished by the tail member)
    # under a caller's dispatch order, one throughput launch from scratch + the tail launch
    order = np.argsort(-ref["info"][:, 4], kind="stable").astype(np.int32)
    with nm.Handle(_cfg(lay, hint, tail_latency=256, staged=-1)) as h:
        h.set_dispatch_order(order)
        r = h.solve(P)
        assert h.last_launch_info()["tail_handed_off"] == 256 and h.last_launch_info()["staged_outer_iterations"] == 0
    _same(r, ref, "dispatch order")
    # with an evaluation budget: the same truncated answers whoever finishes the instance
    with nm.Handle(_cfg(lay, hint, tail_latency=-1, staged=1, max_evaluations=300)) as h:
        b0 = h.solve(P)
    with nm.Handle(_cfg(lay, hint, tail_latency=1024, staged=1, max_evaluations=300)) as h:
        b1 = h.solve(P)
    _same(b1, b0, "budget")
    assert (b0["status"] == 2).sum() > 100


def test_tail_handoff_on_the_general_member_and_in_small_or_other_launches():
    """Rotated ellipses send a call to the general member of the kernel pair (axis_aligned = 0: decided on the device): the tail
    member has its general twin. Batches that do not fill the device several times over, launches without a dispatch order,
    the latency / cooperative families and fp64 run as before."""
    lay = nm.scenarios.ParamLayout(20, 10, 10, 40)
    B = 16384
    P = nm.scenarios.make_batch_chunked(B, lay, seed=43, n_ped=4, n_hyp=10, ped_mode="passing", dtype=np.float32)
    rows = P[:, lay.od:lay.od + 40 * 21 * 6].reshape(B, 40, 21, 6)
    rows[::7, 3, :, 4] = 0.4
    rows[::7, 3, :, 2] *= 1.3                       # (a rotated circle would still be axis-aligned)
    res = {}
    for thr in (-1, 1024):
        with nm.Handle(_cfg(lay, 40, tail_latency=thr, staged=1)) as h:
            res[thr] = h.solve(P)
            li = h.last_launch_info()
            assert li["axis_aligned"] == 2 and li["tail_handed_off"] == max(thr, 0), li
    _same(res[1024], res[-1], "general member")
    assert (res[1024]["info"][:, 7] > 0).sum() > 0
    with nm.Handle(_cfg(lay, 40, tail_latency=200, staged=1)) as h:
        h.solve(P[:900])
        assert h.last_launch_info()["tail_handed_off"] == 0         # B < 5 x the threshold
        h.solve(P[:4096].astype(np.float64), dtype=np.float64)
        assert h.last_launch_info()["tail_handed_off"] == 0         # fp64: no tail member
    with nm.Handle(_cfg(lay, 40, tail_latency=200, staged=-1)) as h:
        h.solve(P)
        assert h.last_launch_info()["tail_handed_off"] == 0         # one launch in index order: nothing ranks the instances
    cfg = _cfg(lay, 40, tail_latency=200)
    cfg.latency_waves = 4
    with nm.Handle(cfg) as h:
        h.solve(P[:2048])
        assert h.last_launch_info()["family"] == "latency" and h.last_launch_info()["tail_handed_off"] == 0
    with pytest.raises(nm.NmpcError):
        nm.Handle(_cfg(lay, 40, tail_latency=-2))
