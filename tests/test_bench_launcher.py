"""`python bench.py --gpus N` must start N ranks by itself (the driver's SCALE command shape), without the launcher
touching the GPU.  Driven here with the bench's CPU stub (gloo): the launcher, the barrier / max-over-ranks timing and
the table gather are the real code paths; only the per-rank step is replaced."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout           # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    rec = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--tiles", "5", "--stub"])
    assert rec["stub"] is True and rec["n_gpus"] == 2 and rec["steps"] == 3
    assert rec["rows"] == 2 * 5 * 3 and rec["row_order_ok"]       # weak scaling: 5 tiles per rank, rows in (tile, label) order
    assert [r["rank"] for r in rec["per_rank"]] == [0, 1] and all(r["rows"] == 15 for r in rec["per_rank"])   # per-rank detail in the one line


def test_launcher_refuses_to_start_ranks_from_a_profiled_process():
    """rocprofv3 preloads its tool into the process (the GPU is initialised before main() runs); child ranks started from there
    would be an exec out of a GPU-initialised process.  The launcher refuses: profiling is single-rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ROCPROFILER_REGISTER_FORCE_LOAD"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--tiles", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "profiled process" in p.stderr


def test_single_rank_needs_no_launcher():
    rec = _run(["--gpus", "1", "--steps", "2", "--tiles", "4", "--stub"])
    assert rec["n_gpus"] == 1 and rec["rows"] == 12 and rec["row_order_ok"]


def test_launcher_reports_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--tiles", "-7"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_a_failed_parity_gate_costs_the_value_and_the_exit_code():
    """a timed configuration counts only if its features match.  A deliberately wrong table (test hook) must null the
    value and end the process -- every rank, hence the launcher -- with code 4."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NYX_BENCH_BREAK_GATE"] = "1"
    for gpus in ("1", "2"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "2", "--tiles", "4", "--stub"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
        rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert rec["value"] is None and "parity gate failed" in rec["error"]


def test_apply_gates_nulls_every_failed_leg():
    sys.path.insert(0, ROOT)
    import bench
    rec = {"value": 1.0, "config": {"parity_check": "tiles 0 vs oracle: ok/ok/ok; all rows: ok"},
           "config4": {"value": 2.0, "ms_per_step": 3.0, "parity_check": "12 MISMATCHES"},
           "gray_depth_64": {"value": 5.0, "parity_check": "ok (rows of the last tile vs oracle)"},
           "intensity_range": {"rows": [{"ns_per_roi": 1.0, "rois_per_s": 2.0, "parity_check": "ok"}, {"ns_per_roi": 1.0, "rois_per_s": 2.0, "parity_check": "3 MISMATCHES"}]},
           "tile_path": {"value": 7.0, "parity_check": None, "irregular": {"value": 8.0, "parity_check": "ROW MISMATCH (labels of the last tile)"}}}
    assert bench.apply_gates(rec) == 4
    assert rec["value"] == 1.0 and rec["config4"]["value"] is None and rec["config4"]["ms_per_step"] is None and rec["gray_depth_64"]["value"] == 5.0
    assert rec["intensity_range"]["rows"][0]["ns_per_roi"] == 1.0 and rec["intensity_range"]["rows"][1]["ns_per_roi"] is None
    assert rec["tile_path"]["value"] == 7.0 and rec["tile_path"]["irregular"]["value"] is None
    assert "config4" in rec["error"] and "tile_path.irregular" in rec["error"]
    ok = {"value": 1.0, "config": {"parity_check": "ok"}, "config4": {"value": 2.0, "parity_check": "ok"}}
    assert bench.apply_gates(ok) == 0 and "error" not in ok
    head = {"value": 1.0, "config": {"parity_check": "FAILED: tiles ...: 5 MISMATCHES"}}
    assert bench.apply_gates(head) == 4 and head["value"] is None
