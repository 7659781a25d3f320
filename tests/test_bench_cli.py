"""bench.py's launcher behaviour: `--gpus N` starts N ranks itself or exits non-zero -- never a smaller run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=900, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_more_gpus_than_visible_is_refused():
    """No launcher and fewer devices than --gpus: non-zero exit, no JSON line (here: no GPU at all; on the GPU box: 1)."""
    r = _run(["--gpus", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


@pytest.mark.gpu
def test_self_spawned_ranks_on_one_device():
    """`--gpus 2` without a launcher: bench.py starts both ranks (here sharing the one GPU, collective over gloo
    because RCCL refuses two ranks on one device) and rank 0 reports n_gpus = 2 and twice the elements."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--workload", "cfg1", "--steps", "6", "--warmup", "2",
              "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_elements"] == 2 * out["config"]["elements_per_gpu"]
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    assert 0 < out["per_gpu_efficiency"] and out["rank0_shard_alone_ms_per_step"] > 0
    # the line proves its own sums: the two ranks hold DIFFERENT slices, the reduced [ds, db] of the last timed step equals their
    # contributions gathered over an independent transport (no reference digests for a 2-rank config 1: parity n/a)
    cv = out["config"]["collective_verified"]
    assert cv["ok"] and cv["ranks"] == 2 and cv["distinct_data_per_rank"] and cv["max_err_over_tol"] <= 1.0, cv
    assert out["parity_vs_reference"]["ok"] is None


@pytest.mark.gpu
def test_a_wrong_sum_prints_the_line_with_ok_false():
    """a corrupted reduction (test switch: the reduced sums scaled by 1 + 1e-3 before they are checked) must not cost the line --
    it comes out with collective_verified.ok = false"""
    r = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--workload", "cfg1", "--steps", "4", "--warmup", "1",
              "--no-cpu-baseline"], extra_env={"LSQ_BENCH_CORRUPT_REDUCED": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cv = out["config"]["collective_verified"]
    assert out["value"] > 0 and cv["ok"] is False and cv["max_err_over_tol"] > 1e3, cv


@pytest.mark.gpu
def test_default_multi_rank_line_carries_the_strong_scaled_config_too():
    """the N > 1 default line (weak-scaled config 2 per rank) also reports BASELINE config 4, [1024,1024,14,14] split over the
    ranks, with its own per-GPU efficiency; here 2 ranks share the one GPU over gloo (the control flow, not the numbers)"""
    r = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "cfg2" in out["config"]["workload"]
    s4 = out["strong_scaled"]
    assert s4["scaling"] == "strong" and s4["global_elements"] == 1024 * 1024 * 14 * 14 and s4["elements_per_gpu"] * 2 == s4["global_elements"]
    assert s4["value"] > 0 and 0 < s4["per_gpu_efficiency"] and 0 < out["per_gpu_efficiency"]
    # ... and over the other routes of the collective (over gloo every route ends in torch.distributed: the control flow)
    assert sorted(s4["other_routes"]) == ["c10d", "native-inline"] and all(r["ms_per_step"] > 0 for r in s4["other_routes"].values())
    # both records verify themselves: reduced == gathered contributions, and against the REFERENCE's outputs for every rank's
    # slice (tests/golden/shard_digests.json: weak-scaled config 2 and config 4 at 2 ranks) -- y / dx by sha256, sums to 1e-6
    for rec, cfg in ((out, out["config"]), (s4, s4)):
        assert cfg["collective_verified"]["ok"] and cfg["collective_verified"]["ranks"] == 2, cfg["collective_verified"]
        par = rec["parity_vs_reference"]
        assert par["ok"] and par["y_dx_sha256_all_ranks_match_reference_slices"] and par["reduced_within_1e-6_sum_abs_terms"], par
        assert par["ranks_checked"] == 2 and abs(par["reduced"][0] - par["reference"][0]) <= par["budget_1e-6_sum_abs_terms"][0]
    assert all(r["collective_verified"] and r["parity_vs_reference"] for r in s4["other_routes"].values()), s4["other_routes"]


def test_default_line_has_secondary_records_declared():
    """(CPU: the flag surface only) the default run appends `secondary` records for the other single-GPU BASELINE configs"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.SECONDARY == ("cfg1", "cfg3", "cfg5", "cfg5_bf16")
    a = b.parse_args([])
    assert a.workload == "cfg2" and a.gpus == 1 and not a.no_secondary
    # the sharded step's collective: the library's own RCCL communicator by default, the record of the default run declared
    assert a.collective == "native" and not a.assume_peers
    assert b.parse_args(["--collective", "native-inline"]).collective == "native-inline"
    assert b.parse_args(["--workload", "cfg4_shard", "--assume-peers", "--collective", "c10d"]).assume_peers
    names = [extra.get("name", w) for w, extra in b.SECONDARY_RUNS]
    assert "cfg4_shard" in names and names.index("cfg4_shard_collective") == names.index("cfg4_shard") + 1


def _timing_bounds(out, sec):
    """the loose timing bounds of the default line (a shared box: the host CPU serves other tenants' jobs too, and the sharded
    step with its collective has the most host time per step of all records -- 40-60 us -- so it is the first to go host-bound)"""
    py, col = out["config"]["python_ctypes_host_layer"], sec["cfg4_shard_collective"]
    problems = []
    if not py["value"] > 0.9 * out["value"]:
        problems.append(("ctypes layer", py["value"], out["value"]))
    if not 0.85 < py["ctypes_over_native"] < 1.15:
        problems.append(("ctypes / native", py["ctypes_over_native"]))
    # typically 1.05-1.07 x the solo step for `native`, 1.02-1.04 x in stream order, 1.2 x through torch.distributed
    if not col["ms_per_step"] <= 1.25 * sec["cfg4_shard"]["ms_per_step"]:
        problems.append(("collective step / solo", col["ms_per_step"], sec["cfg4_shard"]["ms_per_step"], col.get("communicator")))
    if not col["ms_per_step"] <= 1.05 * col["ms_per_step_c10d"]:
        problems.append(("native / c10d", col["ms_per_step"], col["ms_per_step_c10d"]))
    if not col["ms_per_step_native_inline"] <= 1.15 * sec["cfg4_shard"]["ms_per_step"]:
        problems.append(("inline / solo", col["ms_per_step_native_inline"], sec["cfg4_shard"]["ms_per_step"]))
    if not sec["cfg3_x50_foreach"]["step_frac"] > sec["cfg3"]["step_frac"]:
        problems.append(("foreach", sec["cfg3_x50_foreach"]["step_frac"], sec["cfg3"]["step_frac"]))
    if not sec["cfg1_graph"]["ms_per_step"] < sec["cfg1"]["ms_per_step"]:
        problems.append(("graph", sec["cfg1_graph"]["ms_per_step"], sec["cfg1"]["ms_per_step"]))
    return problems


@pytest.mark.gpu
def test_default_line_carries_the_per_channel_half():
    args = ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-measure-traffic", "--secondary-steps", "20"]
    r = _run(args)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    sec = {s["workload"]: s for s in out["secondary"]}
    assert sorted(sec) == ["cfg1", "cfg1_graph", "cfg2_sustained", "cfg3", "cfg3_graph", "cfg3_x50_foreach", "cfg4_shard",
                           "cfg4_shard_collective", "cfg5", "cfg5_bf16"]
    for name, s in sec.items():
        assert "error" not in s and s["value"] > 0 and 0 < s["step_frac"] < 1, s
        assert s["launch"] == ("graph" if name.endswith("_graph") else "eager")
    # the headline next to itself for >= 1 s (events on 1 % of the steps): both figures in `roofline`, both above the target
    sus = out["roofline"]["sustained"]
    assert sec["cfg2_sustained"]["wall_s"] >= 1.0 and sus["steps"] >= 300 and sus["launches_timed"] >= 3, sus
    assert sus["step_frac"] >= 0.70 and out["roofline"]["step_frac"] >= 0.70, (sus, out["roofline"]["step_frac"])
    # the sharded step's record proves its own sums (a world of one: this rank's contribution) against the reference's shard 0 of 8
    assert sec["cfg4_shard_collective"]["collective_verified"]["ok"] and sec["cfg4_shard_collective"]["parity_vs_reference"]["ok"]
    assert "ms_per_step_ctypes_binding" in sec["cfg1"] and sec["cfg5_bf16"]["storage"] == "bfloat16"
    # the headline workload through the Python / ctypes host layer north_star describes, next to the C++ binding's figure,
    # measured on the timed region's own buffers with the per-op split
    py = out["config"]["python_ctypes_host_layer"]
    assert out["config"]["host_binding"] == "native" and py["fwd_ms"] > 0 and py["bwd_ms"] > 0, py
    # one rank's config-4 step WITH its collective (RCCL world of one told it has a peer), over the library's own communicator
    col = sec["cfg4_shard_collective"]
    assert col["collective"] == "native" and col["c10d_route_was"] == "c10d" and col["shape"] == [128, 1024, 14, 14], col
    host = col["host_us_per_step"]
    assert min(host["shard_step_alone"], host["with_native_collective"], host["with_c10d_collective"]) > 0, host
    # BASELINE config 4's per-GPU shard: the step one rank of the 8-GPU job runs, the denominator of the 0.9x target
    assert sec["cfg4_shard"]["shape"] == [128, 1024, 14, 14] and "what" in sec["cfg4_shard"]
    # timing relations (20-step blocks on a shared box): one more run before they count
    problems = _timing_bounds(out, sec)
    if problems:
        r2 = _run(args)
        assert r2.returncode == 0, r2.stderr[-2000:]
        out2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
        again = _timing_bounds(out2, {s["workload"]: s for s in out2["secondary"]})
        assert not again, ("twice in a row", problems, again)


@pytest.mark.gpu
def test_eight_ranks_control_flow_on_one_device():
    """`--gpus 8` end to end without an 8-GPU node: eight self-spawned ranks share the one GPU, collectives over gloo -- rank
    0 alone prints, the weak line and the strong-scaled config 4 ([1024,1024,14,14] / 8 = [128,1024,14,14] per rank) both
    come out.  cfg1-sized weak shards keep eight processes within the box's memory; config 4's shards are the real size."""
    r = _run(["--gpus", "8", "--backend", "gloo", "--single-device", "--workload", "cfg4", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline"], timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["parallelism"] == "dp8"
    assert out["config"]["elements_per_gpu"] == 128 * 1024 * 14 * 14 and out["config"]["global_elements"] == 1024 * 1024 * 14 * 14
    assert out["value"] > 0 and 0 < out["per_gpu_efficiency"]
    # eight TRUE slices of [1024,1024,14,14]: every rank's y / dx match the reference's slice by sha256, every rank's contribution
    # and the reduced sums sit within 1e-6 sum|terms| of the reference CPU csrc's d_scale / d_shift on the whole tensor
    cv, par = out["config"]["collective_verified"], out["parity_vs_reference"]
    assert cv["ok"] and cv["ranks"] == 8 and cv["distinct_data_per_rank"], cv
    assert par["ok"] and par["ranks_checked"] == 8 and par["y_dx_sha256_all_ranks_match_reference_slices"], par
    assert par["each_rank_contribution_within_1e-6_sum_abs_terms"] and par["reduced_within_1e-6_sum_abs_terms"], par


@pytest.mark.gpu
def test_a_failing_rank_shows_its_traceback():
    r = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--workload", "cfg1", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--fail-rank", "1"])
    assert r.returncode != 0
    assert "rank 1 exited with code" in r.stderr and "deliberate failure of rank 1" in r.stderr and "Traceback" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg3", "cfg5_bf16"])
def test_per_channel_workloads_report_their_kernel(workload):
    r = _run(["--workload", workload, "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-measure-traffic"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "per-channel" in out["config"]["workload"]
    assert ("bwd_seg_kernel" if workload == "cfg3" else "bwd_pc_kernel") in out["roofline"]["kernel"]
    assert out["roofline"]["traffic_source"]


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["ok", "wrong", "hung"])
def test_native_route_is_checked_before_anything_is_timed(how):
    """The library's communicator meets real peers for the first time in a driver run: before the timed region it adds up a
    known vector under a deadline and the ranks agree on the outcome; a failure sends the whole run through torch.distributed
    and the record says why -- never a hang, never a lost line.  Here in an RCCL world of one told it has a peer; the failures
    are simulated (an env switch; a negative deadline, with which the communicator is treated as hung: not torn down,
    the process leaves through os._exit once the line is out)."""
    env = {"ok": {}, "wrong": {"LSQ_BENCH_PREFLIGHT_FAIL": "1"}, "hung": {"LSQ_BENCH_PREFLIGHT_S": "-1"}}[how]
    r = _run(["--workload", "cfg4_shard", "--assume-peers", "--steps", "60", "--warmup", "20", "--no-secondary", "--no-cpu-baseline"],
             extra_env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-500:]
    out = json.loads(lines[0])
    pre = out["config"]["collective_preflight"]
    assert out["value"] > 0
    cv, par = out["config"]["collective_verified"], out["parity_vs_reference"]
    assert cv["ok"] and cv["ranks"] == 1, cv                 # whatever the route: the last step's sums are this rank's own
    assert par["ok"] and par["y_dx_sha256_all_ranks_match_reference_slices"] and par["each_rank_contribution_within_1e-6_sum_abs_terms"], par
    if how == "ok":
        assert pre["ok"] and pre["route"] == "native" and out["config"]["collective"].startswith("native")
        assert pre["events"].startswith("no system-scope fence") and "TIMED route" in pre["checked"], pre
        assert out["config"]["communicator"]["side_stream_choice"] >= 2 and out["config"]["communicator"]["checked"]["world"] == 1
    else:
        assert not pre["ok"] and pre["route"] == "c10d" and out["config"]["collective"].startswith("c10d") and pre["why"]
        assert pre["hung"] == (how == "hung")
