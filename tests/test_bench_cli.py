"""bench.py's launcher behaviour without a GPU: `python bench.py --gpus N` from a bare shell re-executes itself under
torch.distributed.run (one rank per GPU), every rank gets as far as the engine's "no HIP device" check and says so, and
the process ends with a non-zero status instead of hanging or printing a number."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def has_gpu():
    from rust_robotics_amd import _ffi

    return int(_ffi.lib().rr_device_count()) > 0


@pytest.mark.skipif(has_gpu(), reason="exercises the no-device path")
def test_self_launch_reaches_the_device_check():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "no launcher detected" in r.stderr and "torch.distributed.run" in r.stderr
    assert r.stderr.count("no HIP device available for rank") == 2, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no JSON line may be printed without a measurement"


@pytest.mark.skipif(has_gpu(), reason="exercises the no-device path")
def test_single_gpu_run_fails_loudly_without_a_device():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no HIP device available" in r.stderr


def _run_snippet(code, env_extra=None, timeout=60):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_stdout_carries_the_json_line_and_nothing_else():
    """gloo / RCCL write banners to file descriptor 1 from C++: after claim_stdout() those land on stderr and the one
    line emit() prints is all there is on stdout (and a world-2 self-launch prints nothing there without a measurement)."""
    r = _run_snippet("import os, bench\n"
                     "bench.claim_stdout()\n"
                     "os.write(1, b'[Gloo] Rank 0 is connected to 1 peer ranks\\n')\n"
                     "print('a library that prints')\n"
                     "bench.emit({'metric': 'm', 'value': 1.5})\n")
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": "m", "value": 1.5}\n'
    assert "[Gloo] Rank 0" in r.stderr and "a library that prints" in r.stderr


def test_deadline_prints_the_headline_and_leaves():
    """a stalled extra leg (or a dead rank) costs at most RR_BENCH_DEADLINE_S: rank 0 prints the headline leg it already
    has, flagged, and exits 0; a rank without a finished headline exits non-zero and prints nothing."""
    code = ("import time, bench\n"
            "bench.claim_stdout()\n"
            "bench._OUT['partial'] = %s\n"
            "bench.start_deadline(0)\n"
            "time.sleep(30)\n")
    r = _run_snippet(code % "{'metric': 'm', 'value': 2.0}", {"RR_BENCH_DEADLINE_S": "0.5"})
    assert r.returncode == 0, r.stderr
    import json

    line = json.loads(r.stdout)
    assert line == {"metric": "m", "value": 2.0, "deadline_exceeded": True}
    assert "deadline of" in r.stderr
    r = _run_snippet(code % "None", {"RR_BENCH_DEADLINE_S": "0.5"})
    assert r.returncode != 0 and r.stdout == ""


def test_the_last_line_is_short_enough_for_the_driver_and_carries_the_contract(tmp_path):
    """Round 4's line was 24.7 KB and the driver's record came back `parsed: null`.  The last stdout line is now a compact
    one (< 4 KB) with the contract's fields, `roofline` and `cpu_baseline`; the full legs are earlier lines and a file.
    Input: round 4's own full record (profiles/r04g_bench_driver_command.json)."""
    import json

    legs_file = tmp_path / "legs.json"
    r = _run_snippet("import json, bench\n"
                     "bench.claim_stdout()\n"
                     "bench.emit(json.load(open('profiles/r04g_bench_driver_command.json')))\n", {"RR_BENCH_LEGS_FILE": str(legs_file)})
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    last = lines[-1]
    assert len(last) < 4096, len(last)
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("fixed-N MCL (BASELINE.json configs[1])")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-5
    full = json.load(open(legs_file))
    assert abs(line["value"] / full["value"] - 1) < 1e-8 and abs(line["ms_per_step"] / full["ms_per_step"] - 1) < 1e-8
    assert line["legs"]["fastslam"][0] == pytest.approx(full["fastslam"]["ms_per_step"], rel=1e-4)
    # every earlier line is a leg in full, each of them JSON on its own
    named = [json.loads(l)["leg"] for l in lines[:-1]]
    assert named[0] == "headline" and "fastslam" in named and "mcl_multinomial" in named


def test_an_oversized_record_still_yields_a_parseable_line():
    import json

    import bench

    big = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
           "config": {"workload": "w" * 5000}, "cpu_baseline": {"value": 1.0, "cores": 1, "kind": "port", "sample": "s" * 9000}}
    for i in range(400):
        big[f"leg{i}"] = {"ms_per_step": 0.1 * i, "roofline": {"frac": 0.5}}
    s = bench.compact_line(big)
    assert len(s) <= bench.LINE_LIMIT and json.loads(s)["value"] == 1.0


def test_the_line_says_how_the_resample_indices_compare_with_the_literal_walk():
    """north_star: "bit-exact resample indices".  Against the literal FLOAT walk on identical weights and draws a handful of 10^6 slots
    pick the neighbouring particle (the walk's own rounding); the line keeps saying how many, per scheme."""
    import json

    import bench

    legs = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r06zzz_bench_default.json")).read().strip().splitlines()]
    out = {k: v for k, v in next(l for l in legs if l.get("leg") == "headline").items() if k != "leg"}
    out["mcl_multinomial"] = {k: v for k, v in next(l for l in legs if l.get("leg") == "mcl_multinomial").items() if k != "leg"}
    line = json.loads(bench.compact_line(out))
    assert line["index_parity"] == {"systematic": [0, 1000000], "multinomial": [3, 1000000]}
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
