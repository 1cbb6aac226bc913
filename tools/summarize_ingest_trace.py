#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --memory-copy-trace of tools/bench_ingest.py -> profiles/<round>_ingest_trace_summary.json.

For the LAST streamed pass in the trace (the runs of host-to-device copies separated by more than 50 ms are passes): the span from
its first copy to its last lec_* kernel, how much of that span the copy engine was busy, the copy rate while busy, the largest gap
between consecutive copies, and for how long copies and lec_* kernels ran AT THE SAME TIME (the overlap the pipeline exists for).

Usage: tools/summarize_ingest_trace.py <trace dir> <out json> [<bench_ingest json line file> [<passes the command ran>]]    (this rocprofv3 records
no copy sizes: the bytes of a pass come from the bench line's bytes_moved)
"""
import csv
import glob
import json
import os
import sys


def read(pattern, d):
    files = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not files:
        raise SystemExit(f"no {pattern} under {d}")
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    d, out = sys.argv[1], sys.argv[2]
    cop = [r for r in read("*memory_copy_trace.csv", d) if "HOST_TO_DEVICE" in r.get("Direction", "") or "H2D" in r.get("Direction", "").upper()]
    ker = [r for r in read("*kernel_trace.csv", d) if "lec_" in r["Kernel_Name"]]
    c = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Size", 0) or 0)) for r in cop)
    big = [x for x in c if x[1] - x[0] >= 500_000]            # the field uploads take milliseconds (tables and coefficients: microseconds)
    passes, cur = [], [big[0]]
    for x in big[1:]:
        if x[0] - cur[-1][1] > 50_000_000:
            passes.append(cur)
            cur = [x]
        else:
            cur.append(x)
    passes.append(cur)
    if len(sys.argv) > 4:                                     # the number of passes the command ran (warm-up + repeats): equal shares of the copies
        n = int(sys.argv[4])
        per = len(big) // n
        passes = [big[i * per:(i + 1) * per] for i in range(n)]
    last = passes[-1]
    t0 = last[0][0]
    k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in ker)
    kin = [x for x in k if x[0] >= t0 and x[0] <= last[-1][1] + 100_000_000]
    t1 = max(last[-1][1], max(x[1] for x in kin))
    busy = sum(e - s for s, e, _ in last)
    nbytes = sum(b for _, _, b in last)
    if nbytes == 0 and len(sys.argv) > 3:
        line = [ln for ln in open(sys.argv[3]) if ln.startswith("{")][-1]
        nbytes = json.loads(line)["bytes_moved"]
    gaps = [b[0] - a[1] for a, b in zip(last, last[1:])]
    # time during which a copy and a lec_* kernel are both running (both lists are sorted and non-overlapping within themselves)
    both, j = 0, 0
    for s, e, _ in last:
        while j < len(kin) and kin[j][1] <= s:
            j += 1
        i = j
        while i < len(kin) and kin[i][0] < e:
            both += max(0, min(e, kin[i][1]) - max(s, kin[i][0]))
            i += 1
    ktime = sum(e - s for s, e, _ in kin)
    res = {"passes_in_trace": len(passes), "copies_in_last_pass": len(last), "bytes": nbytes, "span_ms": (t1 - t0) / 1e6,
           "copy_busy_ms": busy / 1e6, "copy_engine_busy_fraction_of_span": busy / (t1 - t0), "GBs_while_copying": nbytes / busy,
           "GBs_over_span": nbytes / (t1 - t0), "largest_gap_between_copies_ms": max(gaps) / 1e6 if gaps else 0.0,
           "sum_of_gaps_ms": sum(g for g in gaps if g > 0) / 1e6, "lec_kernels_in_pass": len(kin), "lec_kernel_time_ms": ktime / 1e6,
           "kernel_time_overlapped_with_copies_ms": both / 1e6, "fraction_of_kernel_time_hidden_behind_copies": both / ktime if ktime else None,
           "kernel_ms_by_name": {}}
    for s, e, n in kin:
        key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        res["kernel_ms_by_name"][key] = res["kernel_ms_by_name"].get(key, 0.0) + (e - s) / 1e6
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
