#!/usr/bin/env python3
"""Golden vectors for the budget / residual columns, made by the REFERENCE's own module.

``src/utils/calc_budget_and_residual.py`` of the reference needs only numpy and pandas, so it imports in the build
container.  This script imports it from /root/reference, runs ``calc_budget_diff`` and
``calc_residuals`` (calc_budget_and_residual.py:32-56,131-154) on three series of the twelve integrated terms and
commits inputs + outputs as data under tests/golden/budgets/:

  uniform   36 six-hourly steps
  uneven    11 steps on an irregular axis (the reference still divides by dt = t[1] - t[0]: np.gradient(X, dt))
  two_step  2 steps (one-sided differences at both ends)

Only this script touches the reference; the fixtures are what travels (tests/test_host_cpu.py reads them).
Run in the build container:  python tools/make_budget_fixtures.py
"""
import importlib.util
import logging
import os

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/utils/calc_budget_and_residual.py"
OUT = os.path.join(ROOT, "tests", "golden", "budgets")
TERMS = ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe"]


def series(kind: str):
    rng = np.random.default_rng({"uniform": 1, "uneven": 2, "two_step": 3}[kind])
    if kind == "uniform":
        hours = 6.0 * np.arange(36)
    elif kind == "uneven":
        hours = np.cumsum(np.array([0, 6, 6, 3, 3, 12, 6, 1, 5, 6, 24], dtype=np.float64))
    else:
        hours = np.array([0.0, 3.0])
    n = hours.size
    cols = {}
    for i, t in enumerate(TERMS):
        scale = 1e5 if i < 4 else 1.0           # energies in J/m2, the rest in W/m2
        cols[t] = scale * (1.0 + 0.3 * np.sin(0.2 * hours + i) + 0.05 * rng.standard_normal(n))
    dates = np.datetime64("2005-08-08T00:00:00") + (hours * 3600).astype("timedelta64[s]")
    return hours * 3600.0, dates, pd.DataFrame(cols)


def main():
    spec = importlib.util.spec_from_file_location("ref_budget", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    log = logging.getLogger("make_budget_fixtures")
    os.makedirs(OUT, exist_ok=True)
    for kind in ("uniform", "uneven", "two_step"):
        time_s, dates, df = series(kind)
        df.insert(0, "time_s", time_s)
        df.to_csv(os.path.join(OUT, f"{kind}_in.csv"), index=False, float_format="%.17g")
        work = df.drop(columns="time_s").copy()
        work = ref.calc_budget_diff(work, dates, log)
        work = ref.calc_residuals(work, log)
        out = work[[c for c in work.columns if c not in TERMS]]
        out.to_csv(os.path.join(OUT, f"{kind}_out.csv"), index=False, float_format="%.17g")
        print(kind, len(df), "steps ->", list(out.columns))


if __name__ == "__main__":
    main()
