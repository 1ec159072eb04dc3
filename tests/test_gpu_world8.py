"""n_ranks = 8 EXECUTED ("the first 8-GPU lease would be the first time that code path runs"): the BASELINE
8-rank configurations -- configs[3] FastSLAM 8 x 125 000 x 200, configs[4] MCL 8 x 2 000 000 x 64 -- and small worst cases as
eight ranks on the ONE device of the gpurun box, every shard's particles (and maps) array_equal to its block of the
full-size unsharded filter, zero give-ups.  Three wirings: eight shards linked inside one process
(tools/world8_one_device.py), eight processes over hipIpc handles (the deployment's wiring; the _gpu_*_p2p_worker.py
scripts), and `bench.py --gpus 8` with RR_BENCH_SHARE_DEVICE=1.  gloo world 8 on CPU: tests/test_sharded_gloo.py,
tests/test_fs1_sharded_gloo.py.

RR_P2P_CU_PARTITION=1 confines every sharer's stream to its own eighth of the CUs, which lets sharers of ANY size take the
lazy window step (k_step_lazy<kSrcWindow> | k_shard_plan_mark | k_push_window: what eight GPUs would run) and each its own
one-launch plan; without it large sharers take the eager step (see rr_pf_shard_step_p2p).  Both are run.

Reference semantics held across the 8 blocks: fastslam1.rs:205-234 (systematic walk), particle_filter.rs:426-439 (normalisation)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "world8_one_device.py")


def run_tool(cases, extra_env=None, timeout=900):
    # RR_P2P_TIMEOUT_MS: eight shards time-share ONE device here; a peer's kernel can sit behind seven others' (and, with eight
    # processes, behind the scheduler's time slices), so the bound of a peer wait is widened from the deployment's 2 s -- every wait
    # stays bounded, and a give-up still fails the test
    env = dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="12", RR_P2P_TIMEOUT_MS="20000")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, TOOL] + list(cases), capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "WORLD8_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert recs and all(q["world"] == 8 and q["equal_to_unsharded"] and not q["ranks_that_gave_up"] for q in recs), recs
    return recs


@pytest.mark.parametrize("partition", ["0", "1"])
def test_eight_shards_in_one_process_small_and_worst_case(partition):
    """8 x 4 100 (and 8 x 49 000, the largest sharers of a whole device that still take the lazy step): fused / mixed with the
    eager step and an accessor in mid-run / either sender of the weight maximum, and the worst-case resamples -- one heavy
    particle in rank 7 (every slot of ranks 0..6 crosses ranks: the whole window is overhang), in rank 0, one at each end."""
    recs = run_tool(["mcl-small", "mcl-heavy", "mcl-lazy-max"], {"RR_P2P_CU_PARTITION": partition})
    assert all(q["lazy_window_step"] for q in recs if q["mode"] != "mixed")  # (rr_pf_p2p_topology: what the last step really took)
    assert {q["wmax_early"] for q in recs} >= {"0", "1"}
    assert all((q["cus_per_shard"] == [32]) == (partition == "1") for q in recs), [q["cus_per_shard"] for q in recs]


def test_eight_multinomial_shards_in_one_process():
    """the reference's own PF / MCL resampler sharded 8 ways over the peer-to-peer transport: 8 x 4 100 and 8 x 250 000 (bench scene)"""
    recs = run_tool(["mcl-multinomial"])
    assert len(recs) == 2 and all(q["mode"] == "multinomial-p2p" for q in recs)


def test_eight_shards_in_one_process_multi_launch_plan():
    """the same with the plan of a shard as separate launches (the form shards beyond 2^20 particles take)"""
    run_tool(["mcl-small", "mcl-heavy"], {"RR_PF_FUSED_PLAN": "0"})


@pytest.mark.parametrize("partition", ["1", "0"])
def test_config5_as_eight_shards_in_one_process(partition):
    """BASELINE configs[4]: MCL 16 000 000 x 64 as 8 x 2 000 000, against the unsharded 16 000 000-particle filter
    (tests/test_gpu_edge_sizes.py::test_config5_full_size_on_one_gpu's filter) -- the bench scene, and a resample that serves
    every slot from rank 7.  partition = 1: the lazy window step at full size."""
    recs = run_tool(["mcl-config5", "mcl-config5-heavy"], {"RR_P2P_CU_PARTITION": partition}, timeout=1500)
    assert all(q["n_global"] == 16_000_000 and q["landmarks"] == 64 for q in recs)
    assert all(q["lazy_window_step"] == (partition == "1") for q in recs)


def test_config4_as_eight_shards_in_one_process():
    """BASELINE configs[3]: FastSLAM 1.0 1 000 000 x 200 as 8 x 125 000 with their maps (19.3 GB sharded + 19.3 GB unsharded on
    the one device), poses, weights and every landmark of every particle equal to
    tests/test_gpu_fs1_parity.py::test_config4_full_size_on_one_gpu's filter; plus small FastSLAM 1.0 / 2.0 worlds of 8."""
    recs = run_tool(["fs1-small", "fs1-config4"], {"RR_P2P_CU_PARTITION": "1"}, timeout=1800)  # (a stream, hence a hardware queue, of its own per shard)
    big = [q for q in recs if q["case"] == "fs1-config4"]
    assert big and big[0]["n_global"] == 1_000_000 and big[0]["landmarks"] == 200 and any(big[0]["gate_fired"])


def launch(script, args, port, nproc=8, extra_env=None, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", script)] + [str(a) for a in args]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", RR_P2P_TIMEOUT_MS="30000")  # (see run_tool)
    env.update(extra_env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("n_local,steps,env", [(8000, 8, {}), (8000, 8, {"RR_P2P_WMAX_EARLY": "1"}),
                                               (2_000_000, 5, {"RR_P2P_CU_PARTITION": "1", "RR_WORKER_PEAKED": "64"}),
                                               (2_000_000, 5, {"RR_WORKER_PEAKED": "64"})])
def test_eight_processes_over_ipc_handles_mcl(n_local, steps, env):
    """one process per shard, all eight on device 0, peers mapped through hipIpc handles exchanged over gloo -- the wiring of
    `bench.py --gpus 8`; at 2 000 000 per rank (configs[4]) with the bench scene"""
    r = launch("_gpu_p2p_worker.py", [n_local, steps], 29741, extra_env=env, timeout=1500)
    assert r.returncode == 0 and r.stdout.count("P2P_OK") == 8, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("n_local,L,steps", [(3000, 7, 8)])
def test_eight_processes_over_ipc_handles_fastslam(n_local, L, steps):
    """the same for FastSLAM 1.0: 8 x 3 000 x 7 from a host-made state.  (configs[3] at full size -- 8 x 125 000 x 200 -- runs as
    eight shards of ONE process in test_config4_as_eight_shards_in_one_process; as eight PROCESSES it is tools/run_world8.sh's
    ipc_fs1_config4 case, see profiles/r06_world8.md)"""
    r = launch("_gpu_fs1_p2p_worker.py", [n_local, steps, L], 29742, timeout=600)
    assert r.returncode == 0 and r.stdout.count("FS1_P2P_OK") == 8, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_eight_ranks_sharing_the_device():
    """`python bench.py --gpus 8` end to end with all eight ranks on device 0 (RR_BENCH_SHARE_DEVICE=1): self-launch, the gloo
    group of 8, the transport ladder -- the peer-to-peer transport validated against the unsharded filter of all 8 x n particles
    ACROSS the eight processes --, the timed region, rank 0's line.  Not a scaling number (one device): the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, RR_BENCH_SHARE_DEVICE="1", RR_BENCH_DEADLINE_S="600", RR_P2P_CU_PARTITION="1", RR_P2P_TIMEOUT_MS="30000",
               RR_BENCH_EXTRA_WARMUP="40")  # (eight processes time-slicing one GPU: ~50 ms a step, and no rate is being measured here)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--particles", "250000",
                        "--no-extra-legs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    try:  # (the ranks' progress and, after a failed validation, what every rank saw: readable after the run)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "test_bench_eight_ranks.stderr.txt"), "w") as f:
            f.write(r.stderr)
    except OSError:
        pass
    assert r.returncode == 0, r.stderr[-4000:]
    lines = r.stdout.splitlines()
    assert all(ln.startswith("{") for ln in lines) and len(lines[-1]) < 4096, r.stdout[:2000]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0 and "deadline_exceeded" not in d
    legs = {q["leg"]: q for q in map(json.loads, lines[:-1])}
    why = [ln for ln in r.stderr.splitlines() if "VALIDATION" in ln or "gave up" in ln or "validat" in ln][-12:]
    assert "peer-to-peer transport validated bit-identical" in legs["headline"]["config"]["sharding"], (legs["headline"]["config"]["sharding"], why)
    assert legs["sharded"]["transport"].startswith("p2p") and not legs["sharded"]["p2p_timed_out"]
    assert "shared" in json.dumps(d).lower(), "a shared-device line must say that it is not a scaling number"
    for k in range(8):
        assert f"[bench rank {k}" in r.stderr
