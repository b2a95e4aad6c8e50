"""bench.py's timed region (CPU only): W warmup steps then exactly K timed steps, twice - cold, then behind the pre-roll - and
what the line reports about it."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**kw):
    a = types.SimpleNamespace(steps=20, warmup=5, preroll_ms=40.0)
    a.__dict__.update(kw)
    return a


def test_timed_region_counts_steps_and_reports_both_regions():
    import bench
    calls, barriers = [], []

    def run_steps(k):
        calls.append(k)
        time.sleep(k * 1e-4)            # 0.1 ms per step

    dt, region = bench.timed_region(_args(), run_steps, lambda: barriers.append(len(calls)), 1, None, None, "cpu")
    # cold region: 5 + 20; pre-roll: ~40 ms / 0.1 ms; reported region: 5 + 20 again
    assert calls[0] == 5 and calls[1] == 20 and calls[3] == 5 and calls[4] == 20 and len(calls) == 5
    assert 10 <= calls[2] <= 450                        # (sleep granularity and a busy host make a step look longer than 0.1 ms)
    assert region["preroll"]["steps"] == calls[2] and region["preroll"]["asked_ms"] == 40.0
    assert region["cold_start"]["unit"] == "iters/s" and region["cold_start"]["ms_per_step"] > 0
    assert 20 * 0.9e-4 <= dt <= 20 * 1e-3               # the K steps of the second region only
    # a barrier on both sides of each timed region and behind the pre-roll
    assert barriers == [1, 2, 3, 4, 5]


def test_timed_region_without_preroll_is_the_old_form():
    import bench
    calls = []
    dt, region = bench.timed_region(_args(preroll_ms=0.0, steps=7, warmup=2), lambda k: calls.append(k), lambda: None, 1, None, None,
                                    "cpu")
    assert calls == [2, 7] and region is None and dt >= 0.0
