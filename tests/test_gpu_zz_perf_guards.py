"""Performance regression guards -- NOT parity.  This file sorts last (`zz`), so under the driver's `pytest -x` every
parity test has already run when a wall-clock ratio is first asserted: a noisy box can cost these guards, never a §8 row.

VERDICT r5 Weak #8 / Next #3: the 10 % guard on the library's form choice used to sit in tests/test_gpu_round5.py in front of
40+ parity tests.  Its oracle comparison stayed there (test_every_form_of_a_table_gives_the_oracles_numbers); the timing
part is here, with a time-based warm-up like bench.py's and 15 samples per form, medians compared."""
import time

import pytest

from tests.test_gpu_round5 import FORM_POINTS, form_point_table

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


WARM_S = 0.3        # seconds of the same apply before any sample (bench.py's WARM_S)
SAMPLES = 15


def _median_ms(torch, plan, X, T, R):
    out = torch.empty((T, R), dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    n = 0
    while n < 3 or time.perf_counter() - t0 < WARM_S:
        plan.apply(X, out=out)
        torch.cuda.synchronize()
        n += 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(SAMPLES)]
    for a, b in ev:
        a.record()
        plan.apply(X, out=out)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


@pytest.mark.parametrize("kind,param", FORM_POINTS)
def test_default_form_is_within_10_percent_of_the_fastest_forced_form(torch_cuda, kind, param):
    """The form the library picks for a table runs within 10 % of the fastest form it could have been forced into, at six
    points either side of the measured crossovers (tools/form_crossover.py)."""
    from climate_toolbox_amd import engine
    from climate_toolbox_amd.engine import DensePlan
    torch = torch_cuda
    rowptr, col, val, G, R, T = form_point_table(kind, param)
    X = engine.synth_field(T, G, seed=31, base=280.0, amp=60.0, dtype="float32")
    times, picked = {}, None
    plans = {}
    for form in (None, "full", "tiles", "entries"):
        plans[form or "auto"] = DensePlan.from_csr(rowptr, col, val, G, R, form=form)
    picked = {0: "full", 1: "tiles", 2: "entries"}[plans["auto"].info["form"]]
    # two interleaved rounds, the lower median of each form: a disturbance that lasts for one measurement (another tenant of the
    # host, a clock step) cannot make one form look slow
    for _ in range(2):
        for name, plan in plans.items():
            ms = _median_ms(torch, plan, X, T, R)
            times[name] = min(times.get(name, ms), ms)
    for plan in plans.values():
        plan.close()
    forced = {k: v for k, v in times.items() if k != "auto"}
    best = min(forced, key=forced.get)
    print("%s %g: picked %s %.3f ms | full %.3f tiles %.3f entries %.3f | fastest %s" % (
        kind, param, picked, times["auto"], forced["full"], forced["tiles"], forced["entries"], best))
    assert times["auto"] <= 1.10 * forced[best], (kind, param, picked, times)
