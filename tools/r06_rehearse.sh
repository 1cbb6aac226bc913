#!/bin/bash
# Runs on the GPU box (via gpurun): the bench lines of BASELINE configs 4 / 5 and the N > 1 code paths on ONE GPU (gloo ranks share
# cuda:0; RCCL with one rank), each JSON line into gpurun_out/r06/<tag>.json.  Usage: tools/r06_rehearse.sh [set]   (set: n1 | gloo | all)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
SET=${1:-all}
run() { tag=$1; lim=$2; shift; shift; echo "== $tag: $*"; timeout -k 10 $lim "$@" > $O/$tag.json 2> $O/$tag.err || { echo "FAILED $tag"; tail -5 $O/$tag.err; return 1; }; python3 - <<PY
import json
d = json.loads([ln for ln in open("$O/$tag.json") if ln.startswith("{")][-1])
c = d["config"]
print("  value %.1f  ms/pass %.3f  n_gpus %d  backend %s  roofline %.3f" % (d["value"], d["ms_per_step"], d["n_gpus"], c["backend"], d["roofline"]["frac"]))
for k in ("value_incl_producer", "speedup_vs_n1", "n1_stale", "n1_key", "gathered_series_ok", "peer_blocks_ok", "series_equals_n1", "moving_layout"):
    if k in c: print("   ", k, c[k])
if "producer_ms" in c: print("    producer_ms", {k: round(v, 3) for k, v in c["producer_ms"].items() if isinstance(v, float)})
if "moving_check" in c: print("    moving_check", c["moving_check"])
for k in ("cpu_baseline", "parity"):
    if k in d: print("   ", k, {a: b for a, b in d[k].items() if a in ("value", "cores", "ranks_waiting_in_the_closing_barrier", "ok", "worst_rel_to_scale", "levels_worst_rel_to_scale", "shard", "steps")})
for name, leg in (c.get("strong_scaling") or {}).items():
    print("    leg", name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in leg.items() if k in ("value", "ms_per_pass", "speedup_vs_n1", "n1_stale", "per_gpu_roofline_frac", "series_equals_n1", "timesteps_global", "value_incl_producer", "leg_wall_s", "results_finite", "peer_blocks_ok")})
PY
}
if [ "$SET" = n1 ] || [ "$SET" = all ]; then
  # the N = 1 values of the strong-scaling configurations (tools/update_n1.py writes them into profiles/strong_scaling_n1.json)
  LEC_DIST_BACKEND=nccl run n1m4096 400 python3 $R/bench.py --force-dist --moving --timesteps-global 4096 --steps 10 --warmup 2 &&
  LEC_DIST_BACKEND=nccl run n1m4096cube 400 python3 $R/bench.py --force-dist --moving --moving-layout cube --timesteps-global 4096 --cpu-baseline quick --steps 10 --warmup 2 &&
  LEC_DIST_BACKEND=nccl run n1m512 300 python3 $R/bench.py --force-dist --moving --timesteps-global 512 --cpu-baseline quick --steps 20 --warmup 3 &&
  run n1f2048 500 python3 $R/bench.py --timesteps-global 2048 --cpu-baseline quick --steps 3 --warmup 1
fi
if [ "$SET" = gloo ] || [ "$SET" = all ]; then
  export LEC_DIST_BACKEND=gloo
  # the default N > 1 line with its two strong-scaling legs (short series: two and four ranks share one GPU here), as the driver runs it
  run g2default 580 python3 $R/bench.py --gpus 2 --timesteps 8 --steps 5 --warmup 2 --leg-timesteps 64,1024 &&
  run g4default 580 python3 $R/bench.py --gpus 4 --timesteps 8 --cpu-baseline quick --steps 5 --warmup 2 --leg-timesteps 64,1024 &&
  run g2m4096 400 python3 $R/bench.py --gpus 2 --moving --timesteps-global 4096 --cpu-baseline quick --steps 10 --warmup 2 &&
  run g4m4096 400 python3 $R/bench.py --gpus 4 --moving --timesteps-global 4096 --cpu-baseline quick --steps 10 --warmup 2 &&
  run g2chunk 300 python3 $R/bench.py --gpus 2 --timesteps-global 12 --chunk 3 --ny 61 --nx 128 --cpu-baseline quick --steps 2 --warmup 1 &&
  run g3mchunk 300 python3 $R/bench.py --gpus 3 --moving --timesteps-global 50 --chunk 7 --cpu-baseline quick --steps 2 --warmup 1
fi
