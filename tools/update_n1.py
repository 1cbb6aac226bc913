#!/usr/bin/env python3
"""Writes the N = 1 values of bench.py's strong-scaling configurations into profiles/strong_scaling_n1.json.

Usage: tools/update_n1.py <file with a one-GPU strong-scaling bench line> ...

Each input is the stdout of `python bench.py --timesteps-global T [--moving] ...` on ONE GPU (n_gpus == 1, scaling "strong").
An entry is keyed by configuration AND series layout (config.n1_key) and carries the digest of the kernel sources it was measured
on (config.csrc_sha): bench.py prints `n1_stale` when the sources have changed since, as it does for the stored counter pass."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOOK = os.path.join(ROOT, "profiles", "strong_scaling_n1.json")


def main(paths):
    try:
        book = json.load(open(BOOK))
    except Exception:
        book = {}
    book = {k: v for k, v in book.items() if isinstance(v, dict)}          # (rounds 2-5 stored bare numbers without layout or digest: dropped)
    book["_note"] = ("N = 1 values (time steps/s) of bench.py's strong-scaling configurations, keyed by configuration and series layout, each with "
                     "the digest of the kernel sources it was measured on; written by tools/update_n1.py from one-GPU bench lines, read by bench.py "
                     "for config.speedup_vs_n1 / n1_stale")
    for p in paths:
        line = [ln for ln in open(p) if ln.lstrip().startswith("{")][-1]
        d = json.loads(line)
        c = d["config"]
        if d["n_gpus"] != 1 or d["scaling"] != "strong" or "n1_key" not in c:
            raise SystemExit(f"{p}: not a one-GPU strong-scaling line")
        book[c["n1_key"]] = {"value": d["value"], "ms_per_pass": d["ms_per_step"], "csrc_sha": c["csrc_sha"], "backend": c["backend"],
                             "source": os.path.relpath(os.path.abspath(p), ROOT), "steps": d["steps"], "warmup": d["warmup"],
                             **({"value_incl_producer": c["value_incl_producer"], "producer_ms": c["producer_ms"]["total"]} if "producer_ms" in c else {})}
        print(c["n1_key"], book[c["n1_key"]])
    json.dump(book, open(BOOK, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
