"""bench.py prints ONE JSON line with the keys the driver contract names (plus `roofline` and `cpu_baseline`), launches its own
ranks for --gpus N, and refuses a --gpus it cannot honour."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--ny", "61", "--nx", "128", "--steps", "2", "--warmup", "1"]


def _bench(extra, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    return r, lines


def test_bench_json_contract():
    r, lines = _bench(["--timesteps", "3", "--cpu-baseline", "quick"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "timesteps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["results_finite"] is True
    assert d["value"] > 0 and abs(d["value"] - 3 * 2 / (d["ms_per_step"] * 2 * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    ct = rf["conversion_terms"]          # BASELINE.json's target configuration rides in the same line
    assert ct["achieved"] > 0 and abs(ct["frac"] - ct["achieved"] / 8000.0) < 1e-12 and ct["algorithmic_bytes_per_launch"] == 4 * 37 * 61 * 128 * 8 * 3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "timesteps/s" and cb["sample"] and cb["host_cpu_count"] >= 1
    # the CPU leg is also the line's parity check: the oracle evaluated steps of the cube the GPU was timed on
    pr = d["parity"]
    assert pr["ok"] is True and pr["worst_rel_to_scale"] <= 1e-9 and pr["levels_worst_rel_to_scale"] <= 1e-9
    assert pr["steps"] == 2 and pr["terms_compared"] == 16 and pr["level_tables_compared"] == 21 and "59 x 128" in pr["box"]     # polar rows left out
    assert d["config"]["baseline_config"]["id"] is None        # a small grid is not a BASELINE configuration (3 needs 37 x 721 x 1440)
    # where a pass's time goes
    sg = d["config"]["segments_ms"]
    for k in ("stage1", "stage2", "gather", "pass_total_synchronised", "pass_timed_unsynchronised", "stage1_kernels_hip_events", "fixed_cost_per_pass"):
        assert k in sg and sg[k] >= 0 or k == "fixed_cost_per_pass", k


def test_bench_starts_its_own_ranks_and_counts_them():
    """`python bench.py --gpus 2` with no launcher environment: the parent starts two rank processes (gloo rendezvous, both on
    this box's one GPU -- RCCL itself needs a GPU per rank) and rank 0 reports n_gpus = 2, twice the global series."""
    r, lines = _bench(["--gpus", "2", "--timesteps", "3", "--cpu-baseline", "none", "--no-strong-legs"] + SMALL, env={"LEC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert "strong_scaling" not in d["config"]
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines          # stdout holds the JSON line and nothing else (library banners: stderr)
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["timesteps_global"] == 6 and d["scaling"] == "weak"
    assert d["config"]["backend"] == "gloo" and d["config"]["results_finite"] is True and d["config"]["gathered_series_ok"] is True
    # the run verifies itself: ranks connected, every rank's block of the gathered series, per-rank clocks and segments
    c = d["config"]
    assert c["rccl_ranks_seen"] == 2 and c["peer_blocks_ok"] is True and c["peer_blocks"] == [True, True] and len(c["rank_devices"]) == 2
    assert len(c["ms_per_step_per_rank"]) == 2 and max(c["ms_per_step_per_rank"]) <= d["ms_per_step"] * (1 + 1e-9)
    assert set(c["segments_ms_min_over_ranks"]) == set(k for k in c["segments_ms"] if k in c["segments_ms_min_over_ranks"])
    assert all(c["segments_ms_min_over_ranks"][k] <= c["segments_ms"][k] + 1e-12 for k in c["segments_ms_min_over_ranks"])
    assert len(c["series_sha256"]) == 64 and c["series_digest_key"] == "fixed_f64_all_37x61x128_T6" and c["series_equals_n1"] is None
    sg = d["config"]["segments_ms"]
    for k in ("stage1", "stage2", "mask_all_reduce", "gather", "gather.staging_d2h", "gather.collective", "gather.staging_h2d", "fixed_cost_per_pass"):
        assert k in sg, k


def test_bench_under_the_drivers_launch_line(tmp_path):
    """The driver's own N > 1 launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- the launcher's environment names the ranks, bench.py starts nobody, rank 0
    prints the one line with both strong-scaling legs in it (gloo here: both ranks share this box's one GPU)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    n1 = str(tmp_path / "n1.json")
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["LEC_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--timesteps", "3", "--ny", "61", "--nx", "128",
           "--cpu-baseline", "quick", "--leg-timesteps", "12,16", "--n1-file", n1]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and c["timesteps_global"] == 6
    assert c["backend"] == "gloo" and c["rccl_ranks_seen"] == 2 and c["peer_blocks_ok"] is True and c["gathered_series_ok"] is True
    assert d["cpu_baseline"]["value"] > 0 and d["parity"]["ok"] is True
    legs = c["strong_scaling"]
    assert set(legs) == {"config4", "config5"} and all(legs[k]["results_finite"] and legs[k]["peer_blocks_ok"] for k in legs)
    assert legs["config4"]["timesteps_global"] == 12 and legs["config5"]["timesteps_global"] == 16 and "producer_ms" in legs["config5"]


def test_a_two_rank_line_carries_the_cpu_leg_and_the_parity_of_rank_0():
    """north_star: the N-GPU throughput "next to the reference CPU path timed on the node's own host cores in the same run".  Rank 0 runs
    the oracle after the timed region while rank 1 waits in the closing barrier; `parity` is rank 0's shard against the oracle."""
    r, lines = _bench(["--gpus", "2", "--timesteps", "3", "--cpu-baseline", "full", "--no-strong-legs"] + SMALL, env={"LEC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines and len(lines) == 1
    d = json.loads(lines[0])
    cb, pr = d["cpu_baseline"], d["parity"]
    assert d["n_gpus"] == 2 and cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["ranks_waiting_in_the_closing_barrier"] == 1
    assert "all_cores" not in cb and "1 repetition" in cb["sample"]                 # "full" becomes the one-thread leg at N > 1
    assert pr["ok"] is True and pr["steps"] == 3 and pr["shard"] == "rank 0 of 2: global time steps 0..2" and pr["terms_compared"] == 16
    assert d["config"]["baseline_config"]["id"] is None


def test_bench_runs_the_rccl_code_path_with_one_rank():
    """--force-dist: process group over backend "nccl" (= RCCL), barrier, the mask all_reduce and the gather run with world size 1 --
    the N > 1 code path of the product on the one GPU this box has (more RCCL ranks need more GPUs)."""
    r, lines = _bench(["--force-dist", "--timesteps", "3", "--cpu-baseline", "quick"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["cpu_baseline"]["value"] > 0 and d["parity"]["ok"] is True          # the CPU leg runs on the distributed code path too
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines          # RCCL's version banner went to stderr
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["results_finite"] is True and d["config"]["gathered_series_ok"] is True
    assert "mask_all_reduce" in d["config"]["segments_ms"] and "gather.collective" in d["config"]["segments_ms"]
    c = d["config"]
    assert c["rccl_ranks_seen"] == 1 and c["devices_distinct"] is True and c["peer_blocks_ok"] is True and c["peer_blocks"] == [True]
    assert c["rank_devices"][0]["device_index"] == 0 and c["rank_devices"][0]["name"]


def test_series_digest_of_a_two_rank_run_equals_the_one_gpu_digest(tmp_path):
    """The synthetic fields are seeded per GLOBAL time step and the kernels are bitwise reproducible under sharding, so the series of
    an N-rank run is the series of the one-GPU run: `--write-digest` stores the N = 1 per-step checksums, and a run finds
    `series_equals_n1` true -- or says which steps differ."""
    book = str(tmp_path / "digests.json")
    r, lines = _bench(["--timesteps-global", "6", "--cpu-baseline", "none", "--digest-file", book, "--write-digest"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    one = json.loads(lines[0])["config"]
    stored = json.load(open(book))["fixed_f64_all_37x61x128_T6"]
    assert stored["sha256"] == one["series_sha256"] and stored["steps"] == 6 and len(stored["per_step"]) == 96
    r, lines = _bench(["--gpus", "2", "--timesteps", "3", "--cpu-baseline", "none", "--no-strong-legs", "--digest-file", book] + SMALL, env={"LEC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    two = json.loads(lines[0])["config"]
    assert two["series_digest_key"] == "fixed_f64_all_37x61x128_T6" and two["series_sha256"] == one["series_sha256"]
    assert two["series_equals_n1"] is True and "series_steps_differing_from_n1" not in two
    # a stored digest that differs in two steps: the line says false and names them
    bk = json.load(open(book))
    ps = bk["fixed_f64_all_37x61x128_T6"]["per_step"]
    bk["fixed_f64_all_37x61x128_T6"].update(sha256="0" * 64, per_step=ps[:16] + "f" * 16 + ps[32:80] + "0" * 16)
    json.dump(bk, open(book, "w"))
    r, lines = _bench(["--gpus", "2", "--timesteps", "3", "--cpu-baseline", "none", "--no-strong-legs", "--digest-file", book] + SMALL, env={"LEC_DIST_BACKEND": "gloo"})
    bad = json.loads(lines[0])["config"]
    assert bad["series_equals_n1"] is False and bad["series_steps_differing_from_n1"] == {"count": 2, "first": [1, 5]}


def test_bench_moving_checks_its_kernel_against_the_independent_one():
    r, lines = _bench(["--moving", "--timesteps", "12", "--cpu-baseline", "none", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    mc = json.loads(lines[0])["config"]["moving_check"]
    assert mc["ok"] is True and mc["steps"] == 8 and mc["timed_pass_bit_identical"] is True and mc["terms_max_rel_diff_vs_row_sweep"] <= 1e-9


def _moving_line_is_whole(d, steps, n_cpu, held):
    """A --moving line says what it times: the producer of the packed series (per-step slice + dT/dt, both inside the reference's
    clock: lorenzcycletoolkit.py:173-199) on events of its own, the rate with it inside, the oracle's moving framework as the CPU
    leg and the parity of the TIMED pass's records against it."""
    c = d["config"]
    assert c["moving_layout"] == "packed" and "OUTSIDE this region" in c["timed_region"] and "producer_ms" in c["timed_region"]
    pm = c["producer_ms"]
    assert pm["pack"] > 0 and pm["dtdt"] > 0 and abs(pm["total"] - pm["pack"] - pm["dtdt"]) < 1e-9
    assert abs(c["ms_per_step_incl_producer"] - d["ms_per_step"] - pm["total"]) < 1e-9
    assert abs(c["value_incl_producer"] - c["timesteps_global"] / (c["ms_per_step_incl_producer"] * 1e-3)) <= 1e-6 * d["value"] and c["value_incl_producer"] < d["value"]
    cb, pr = d["cpu_baseline"], d["parity"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "timesteps/s" and "moving framework" in cb["sample"]
    assert f"{held} time steps" in cb["sample"]
    assert pr["ok"] is True and pr["steps"] == n_cpu and pr["terms_compared"] == 16 and pr["level_tables_compared"] == 21
    assert pr["worst_rel_to_scale"] <= 1e-9 and pr["levels_worst_rel_to_scale"] <= 1e-9 and pr["compared_with"] == "the records of the timed GPU pass"


def test_a_moving_line_times_its_producer_and_carries_the_cpu_leg():
    r, lines = _bench(["--moving", "--timesteps", "12", "--cpu-baseline", "quick", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    _moving_line_is_whole(d, 2, 3, 4)
    assert d["config"]["producer_ms"]["reproduced_series_is_the_timed_one"] is True and "re-produced 3 times" in d["config"]["producer_ms"]["how"]
    # the cube layout slices and differentiates inside the kernel: no producer, and the line says so
    r, lines = _bench(["--moving", "--moving-layout", "cube", "--timesteps", "12", "--cpu-baseline", "quick", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert "producer_ms" not in d["config"] and "both inside this region" in d["config"]["timed_region"]
    assert d["parity"]["ok"] is True and d["cpu_baseline"]["value"] > 0


def test_a_two_rank_moving_line_is_whole_too():
    r, lines = _bench(["--gpus", "2", "--moving", "--timesteps", "6", "--cpu-baseline", "full", "--steps", "2", "--warmup", "1"], env={"LEC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["timesteps_global"] == 12 and "strong_scaling" not in d["config"]
    _moving_line_is_whole(d, 2, 3, 4)            # "full" becomes the 3-step one-thread leg at N > 1
    assert d["cpu_baseline"]["ranks_waiting_in_the_closing_barrier"] == 1 and d["parity"]["shard"] == "rank 0 of 2: global time steps 0..5"


def test_a_forced_rccl_moving_line_is_whole_too():
    r, lines = _bench(["--force-dist", "--moving", "--timesteps", "12", "--cpu-baseline", "quick", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "nccl"
    _moving_line_is_whole(d, 2, 3, 4)


def test_strong_scaling_lines_carry_the_cpu_leg_and_an_honest_n1(tmp_path):
    """A chunked strong-scaling line (BASELINE config 4's shape) has `cpu_baseline` / `parity` from the first chunk's first steps; a one-GPU
    strong line IS the N = 1 value (speedup 1.0); an entry of --n1-file is keyed by the series' layout and stamped with the kernel
    sources' digest, and a line says when that stamp is stale."""
    n1 = str(tmp_path / "n1.json")
    small = ["--ny", "61", "--nx", "128", "--steps", "2", "--warmup", "1", "--n1-file", n1]
    r, lines = _bench(["--timesteps-global", "8", "--chunk", "3", "--cpu-baseline", "quick"] + small)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["scaling"] == "strong" and c["chunk"] == 3 and d["parity"]["ok"] is True and d["parity"]["steps"] == 2 and d["cpu_baseline"]["value"] > 0
    assert "first chunk" in d["parity"]["data"]
    assert c["speedup_vs_n1"] == 1.0 and c["n1_value"] == d["value"] and c["n1_key"] == "fixed_f64_all_T8" and c["n1_stored"] is None and c["n1_stale"] is None
    # two ranks against a stored N = 1 entry: fresh, then stale
    json.dump({"fixed_f64_all_T8": {"value": d["value"], "csrc_sha": c["csrc_sha"], "source": "test"}}, open(n1, "w"))
    r, lines = _bench(["--gpus", "2", "--timesteps-global", "8", "--chunk", "3", "--cpu-baseline", "quick"] + small, env={"LEC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    d2 = json.loads(lines[0])
    c2 = d2["config"]
    assert c2["n1_stale"] is False and abs(c2["speedup_vs_n1"] - d2["value"] / d["value"]) < 1e-12 and c2["n1_stored"]["source"] == "test"
    assert d2["parity"]["ok"] is True and d2["parity"]["shard"] == "rank 0 of 2: global time steps 0..3"
    json.dump({"fixed_f64_all_T8": {"value": d["value"], "csrc_sha": "0" * 16, "source": "test"}}, open(n1, "w"))
    r, lines = _bench(["--timesteps-global", "8", "--chunk", "3", "--cpu-baseline", "none"] + small)
    c3 = json.loads(lines[0])["config"]
    assert c3["n1_stale"] is True and c3["speedup_vs_n1"] == 1.0           # (a one-GPU run is its own N = 1 value whatever is stored)
    # the moving configuration keys its entry by layout
    r, lines = _bench(["--moving", "--timesteps-global", "10", "--chunk", "4", "--cpu-baseline", "quick", "--steps", "2", "--warmup", "1", "--n1-file", n1])
    assert r.returncode == 0, r.stderr[-2000:]
    dm = json.loads(lines[0])
    assert dm["config"]["n1_key"] == "moving_packed_f64_all_T10" and dm["config"]["speedup_vs_n1"] == 1.0
    _moving_line_is_whole(dm, 2, 3, 4)
    assert "inside every timed pass" in dm["config"]["producer_ms"]["how"]


def test_the_default_two_rank_line_carries_the_strong_scaling_legs(tmp_path):
    """`--gpus N` and nothing else is what the driver passes: the line then holds BASELINE configs 4 and 5 as config.strong_scaling
    (here with short series: --leg-timesteps), each with its value, its speed-up over the stored one-GPU value and its self-checks."""
    n1 = str(tmp_path / "n1.json")
    json.dump({"fixed_f64_all_T12": {"value": 100.0, "csrc_sha": "0" * 16, "source": "test"},
               "moving_packed_f64_all_T16": {"value": 1000.0, "value_incl_producer": 500.0, "csrc_sha": "0" * 16, "source": "test"}}, open(n1, "w"))
    r, lines = _bench(["--gpus", "2", "--timesteps", "3", "--cpu-baseline", "quick", "--leg-timesteps", "12,16", "--n1-file", n1] + SMALL, env={"LEC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "weak" and d["config"]["timesteps_global"] == 6 and d["parity"]["ok"] is True      # the headline is what it was
    ss = d["config"]["strong_scaling"]
    for name, T, key in (("config4", 12, "fixed_f64_all_T12"), ("config5", 16, "moving_packed_f64_all_T16")):
        leg = ss[name]
        for k in ("value", "ms_per_pass", "speedup_vs_n1", "n1_stale", "per_gpu_roofline_frac", "series_equals_n1", "results_finite", "peer_blocks_ok"):
            assert k in leg, (name, k)
        assert leg["timesteps_global"] == T and leg["passes"] == 2 and leg["n1_key"] == key and leg["n1_stale"] is True
        assert leg["value"] > 0 and abs(leg["speedup_vs_n1"] - leg["value"] / leg["n1_value"]) < 1e-12
        assert leg["results_finite"] is True and leg["peer_blocks_ok"] is True
    assert ss["config4"]["baseline_config"]["id"] is None and ss["config5"]["baseline_config"]["id"] == 5
    assert ss["config5"]["producer_ms"]["total"] > 0 and ss["config5"]["value_incl_producer"] < ss["config5"]["value"]
    assert abs(ss["config5"]["speedup_vs_n1_incl_producer"] - ss["config5"]["value_incl_producer"] / 500.0) < 1e-12
    assert ss["config5"]["moving_check"]["ok"] is True


def test_bench_refuses_what_it_cannot_launch():
    import torch
    n = torch.cuda.device_count()
    r, lines = _bench(["--gpus", str(n + 1), "--cpu-baseline", "none"] + SMALL)       # RCCL: one GPU per rank
    assert r.returncode != 0 and not lines and "GPU" in r.stderr
    r, lines = _bench(["--gpus", "2", "--cpu-baseline", "none"] + SMALL, env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and not lines and "contradicts WORLD_SIZE" in r.stderr


@pytest.mark.parametrize("extra", [[], ["--moving"]])
def test_bench_strong_scaling_chunked_equals_resident(extra):
    """--timesteps-global: the series is fixed and streamed through HBM in chunks with a one-step T halo; throughput is reported
    with scaling "strong" and the chunked pass gives the resident pass's results (bit-identical records: same kernels)."""
    size = [] if extra else ["--ny", "61", "--nx", "128"]
    out = []
    for chunk in ("3", "0"):
        r, lines = _bench(["--timesteps-global", "8", "--chunk", chunk, "--cpu-baseline", "none", "--steps", "2", "--warmup", "1"] + size + extra)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(json.loads(lines[0]))
    a, b = out
    assert a["scaling"] == b["scaling"] == "strong" and a["config"]["timesteps_global"] == 8 and a["n_gpus"] == 1
    assert a["config"]["chunk"] == 3 and "chunk" not in b["config"] and a["config"]["results_finite"] and b["config"]["results_finite"]
    assert a["value"] > 0 and a["config"]["wall_ms_per_step_incl_generation"] >= a["ms_per_step"]
