#!/bin/bash
# Runs on the GPU box (via gpurun): rehearsals of the N > 1 bench path on ONE GPU (gloo ranks share cuda:0; RCCL with one rank),
# each printing its JSON line into gpurun_out/r05_<tag>.json.  The figures of interest are config.segments_ms, not `value`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
run() { tag=$1; shift; echo "== $tag: $*"; timeout -k 10 240 "$@" > $O/r05_$tag.json 2> $O/r05_$tag.err || { echo "FAILED $tag"; tail -5 $O/r05_$tag.err; return 1; }; python3 - <<PY
import json
d = json.loads([ln for ln in open("$O/r05_$tag.json") if ln.startswith("{")][-1])
print("  value %.1f  ms/pass %.3f  n_gpus %d  backend %s" % (d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["backend"]))
for k, v in sorted(d["config"].get("segments_ms", {}).items()):
    print("    %-34s %9.3f ms" % (k, v))
for k in ("gathered_series_ok", "moving_check", "peer_blocks_ok", "series_equals_n1", "baseline_config", "moving_layout"):
    if k in d["config"]: print("   ", k, d["config"][k])
for k in ("cpu_baseline", "parity"):
    if k in d: print("   ", k, {a: b for a, b in d[k].items() if a in ("value", "cores", "ranks_waiting_in_the_closing_barrier", "ok", "worst_rel_to_scale", "shard", "steps")})
PY
}
export LEC_DIST_BACKEND=gloo
run g1m  python3 $R/bench.py --moving --timesteps-global 1024 --cpu-baseline none --steps 10 --warmup 2 &&
run g2m  python3 $R/bench.py --gpus 2 --moving --timesteps-global 1024 --cpu-baseline none --steps 10 --warmup 2 &&
run g4m  python3 $R/bench.py --gpus 4 --moving --timesteps-global 1024 --cpu-baseline none --steps 10 --warmup 2 &&
run g2m4096 python3 $R/bench.py --gpus 2 --moving --timesteps-global 4096 --cpu-baseline none --steps 10 --warmup 2 &&
run g4m4096 python3 $R/bench.py --gpus 4 --moving --timesteps-global 4096 --cpu-baseline none --steps 10 --warmup 2 &&
run g2f  python3 $R/bench.py --gpus 2 --timesteps 8 --steps 5 --warmup 2 &&      # default CPU leg: rank 0 runs the oracle while rank 1 waits in the closing barrier
run g4f  python3 $R/bench.py --gpus 4 --timesteps 8 --cpu-baseline quick --steps 5 --warmup 2 &&
LEC_DIST_BACKEND=nccl run n1m4096 python3 $R/bench.py --force-dist --moving --timesteps-global 4096 --cpu-baseline none --steps 10 --warmup 2 &&
LEC_DIST_BACKEND=nccl run n1m512 python3 $R/bench.py --force-dist --moving --timesteps-global 512 --cpu-baseline none --steps 20 --warmup 3 &&
LEC_DIST_BACKEND=nccl run n1f  python3 $R/bench.py --force-dist --timesteps 16 --cpu-baseline none --steps 5 --warmup 2
LEC_DIST_BACKEND=gloo run g2chunk python3 $R/bench.py --gpus 2 --timesteps-global 12 --chunk 3 --ny 61 --nx 128 --cpu-baseline none --steps 2 --warmup 1
LEC_DIST_BACKEND=gloo run g3mchunk python3 $R/bench.py --gpus 3 --moving --timesteps-global 50 --chunk 7 --cpu-baseline none --steps 2 --warmup 1
