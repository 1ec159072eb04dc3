"""The one-launch resample plan (k_quantize_plan_mark: workgroups hand their tile sums to each other inside the kernel)
needs all of its workgroups on the device at once.  When the device does not grant that -- another process on the GPU --
the launch must DEGRADE (serial plan in its last workgroup, then the multi-launch plan), never fail and never change a bit."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_gpu_plan_worker.py")


def run(args, timeout=400, **env):
    e = dict(os.environ, PYTHONPATH=ROOT)
    e.update(env)
    return subprocess.run([sys.executable, WORKER] + args, capture_output=True, text=True, timeout=timeout, env=e)


@pytest.mark.parametrize("kind", ["pf", "mcl", "fs"])
def test_serial_plan_is_bit_identical_to_the_multi_launch_plan(kind):
    """RR_PF_PLAN_TIMEOUT_US=0: nobody waits, so the give-up protocol and the serial plan run on an idle device -- a gated
    particle filter (in-step estimates included), an every-step MCL and FastSLAM 1.0 (weights rewritten by the plan)."""
    r = run(["serial", kind], RR_PF_PLAN_TIMEOUT_US="0")
    assert r.returncode == 0 and "PLAN_SERIAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_two_processes_on_one_gpu_do_not_break_each_other():
    """Two processes step a 1e6-particle filter on the same GPU at the same time: each one's one-launch plan holds CU slots
    the other one's workgroups wait for (round 2: both timed out after 1.5 s and latched RR_RUNTIME_ERROR).  Now: no error,
    and both end with exactly the particle set of an undisturbed run."""
    ref = run(["reference"])
    assert ref.returncode == 0 and "REFERENCE" in ref.stdout, (ref.stdout[-1000:], ref.stderr[-3000:])
    want = ref.stdout.split("REFERENCE")[1].split()[0]
    with tempfile.TemporaryDirectory() as box:
        e = dict(os.environ, PYTHONPATH=ROOT)
        ps = [subprocess.Popen([sys.executable, WORKER, "contend", box, str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
              for k in range(2)]
        outs = [p.communicate(timeout=600) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0 and "CONTEND" in so, (so[-1000:], se[-3000:])
        f = so.split("CONTEND")[1].split()
        assert f[1] == want, f"process {f[0]} ended with a different particle set (give-ups {f[2]}, one-launch still on: {f[3]}, {f[4]} s)"
    print("give-ups / seconds:", [(so.split("CONTEND")[1].split()[2], so.split("CONTEND")[1].split()[4]) for so, _ in outs])


def test_one_launch_plan_serves_the_headline_size():
    """1e6 particles = 489 tiles: all of k_quantize_plan_mark's workgroups have to fit the device at once (two 512-thread
    workgroups per CU, i.e. at most 128 VGPRs).  A register-count regression silently sends the headline workload down the
    two-launch plan (round 3 saw exactly that while the kernel was being reworked); this notices."""
    import numpy as np

    import rust_robotics_amd.localization as loc
    from tests import helpers as H

    n, L = 1_000_000, 32
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
    lms = H.landmarks_grid(L, 1)
    rng = np.random.default_rng(2)
    pf.profile_enable(1)
    for t in range(4):
        (pf.step_async_estimate if t % 2 else pf.step_async)([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
    prof = pf.profile_read()
    pf.profile_enable(0)
    assert prof["k_quantize_reduce"][0] == 0, f"the two-launch plan ran: {prof}"
    assert prof["k_cdf"][0] == 4
    assert pf.plan_stats() == (0, True)
