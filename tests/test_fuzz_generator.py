"""tools/fuzz_parity.py draws every random number of a case up front, so that `fuzz_parity.py <s> <seed> <first_case>` can replay
the generator to the case a long soak stopped at (round 5: case 19 899 of seed 101).  That only works if a SKIPPED
case consumes exactly the draws a RUN case does: checked here on the CPU (no GPU call is made by drawing)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(seed):
    argv = sys.argv
    sys.argv = ["fuzz_parity.py", "1", str(seed)]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        if "fuzz_parity" in sys.modules:
            del sys.modules["fuzz_parity"]
        return importlib.import_module("fuzz_parity")
    finally:
        sys.argv = argv
        sys.path.pop(0)


def light(q):
    """a case without its two big arrays, cameras as plain lists"""
    out = {k: v for k, v in q.items() if k not in ("keys", "vals", "cam", "cam2")}
    for c in ("cam", "cam2"):
        out[c] = {k: (np.asarray(v).tolist() if not isinstance(v, (int, float)) else v) for k, v in q[c].items()}
    return out


def test_a_skipped_case_consumes_the_draws_of_a_run_case():
    a, cases_a = load(9), []
    for case in range(1, 13):
        cases_a.append(light(a.draw_case(case, skipped=False)))
    b, cases_b = load(9), []
    for case in range(1, 13):
        cases_b.append(light(b.draw_case(case, skipped=case % 2 == 1)))      # every other case skipped
    assert cases_a == cases_b
    assert any("path" in q for q in cases_a) and any("rect" in q for q in cases_a)
    kinds = {q["kind"] for q in cases_a}
    assert len(kinds) >= 3


def test_the_sort_input_of_a_case_is_a_function_of_the_seed_alone():
    a = load(123)
    q1 = a.draw_case(1, skipped=False)
    b = load(123)
    q2 = b.draw_case(1, skipped=False)
    assert (q1["keys"] == q2["keys"]).all() and (q1["vals"] == q2["vals"]).all()
    assert q1["keys"].dtype == np.uint32 and len(q1["keys"]) == q1["count"]
    assert sorted(q1["vals"].tolist()) == list(range(q1["count"]))
