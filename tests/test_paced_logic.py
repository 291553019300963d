"""The paced loops themselves (wmix_amd/realtime.py), without a GPU: a tick is released on an ABSOLUTE schedule and its latency runs from
the SCHEDULED release -- a tick that starts late because its predecessor overran carries that backlog (src/wmix.c:536-538, 820: the
reference's own loop sleeps `interval - elapsed` and falls behind the same way).  Fake ticks that take a fixed time stand in for the
device."""
import time

import numpy as np


def _busy(ms):
    t = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t:
        pass


def test_latency_runs_from_the_scheduled_release():
    from wmix_amd.realtime import latency_summary, paced_loop
    lat, lag, _ = paced_loop(lambda k: _busy(1.0), 5.0, 40)
    s = latency_summary(lat, lag, 5.0)
    assert s["budget_ms"] == 3.0 and s["misses"] == 0 and 0.9 < s["p50_ms"] < 2.5 and s["release_lag_p50_ms"] < 1.0, s
    # a tick that takes longer than its period: every tick starts where the last one ended, and says so
    lat, lag, _ = paced_loop(lambda k: _busy(7.0), 5.0, 12)
    assert np.all(np.diff(lat) > 1.0) and lat[-1] > 7.0 + 2.0 * 10 and lag[-1] > 2.0 * 10
    s = latency_summary(lat, lag, 5.0)
    assert s["misses"] == 12 and s["overruns_of_the_period"] == 12 and 6.9 < s["service_p50_ms"] < 8.0


def test_one_slow_tick_is_paid_by_its_successors_only_while_the_backlog_lasts():
    from wmix_amd.realtime import paced_loop
    lat, lag, _ = paced_loop(lambda k: _busy(12.0 if k == 5 else 1.0), 5.0, 14)
    assert lat[4] < 3.0 and lat[5] > 11.9 and lag[6] > 6.5 and lat[6] > 7.5  # tick 6 was due 5 ms into tick 5's 12
    assert lat[7] > 3.0 and lat[9] < 3.0                                       # 1 ms ticks drain the backlog 4 ms per period


def test_staggered_groups_complete_out_of_band():
    """paced_groups: group g of P is released tick / P after group g - 1; submit returns at once, the completion is seen by polling"""
    from wmix_amd.realtime import paced_groups
    P, tick = 4, 20.0
    done_at = {}

    def submit(g):
        done_at[g] = time.perf_counter() + 3e-3  # the "device" needs 3 ms per group

    def poll(g):
        return time.perf_counter() >= done_at[g]

    def wait(g):
        while not poll(g):
            pass
    seen = []
    lat, lag, _ = paced_groups(submit, poll, wait, P, tick, 6, after=lambda j, g: seen.append((j, g)))
    assert lat.size == 24 and np.all(lat > 2.9) and np.median(lat) < 4.5 and np.median(lag) < 1.0
    assert sorted(seen) == [(j, j % P) for j in range(24)]
