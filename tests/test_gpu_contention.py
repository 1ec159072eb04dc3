"""Eight processes on one device, each creating the SAME unsharded filter and stepping it at once -- after trying (and failing, on a shared
device) to set up the collective transports, as bench.py's ladder does -- must end with the same particles in every process, round after
round (tools/contention_soak.py).  This is the recipe that reproduces round 6's create-time race with the library from before the fix
(a plain hipMemset of the resample markers, which returns before it has run: 1 round of 8 differed, profiles/r06y_contention_soak_old_vs_new.txt)
and has not failed since (rr::memset_on).  Poisoned allocations make "an earlier tenant's bytes" certain."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("poison", ["0", "0x3f"])
def test_same_filter_in_eight_processes_at_once(poison):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, RR_P2P_CU_PARTITION="1", RR_P2P_TIMEOUT_MS="30000", RR_DEBUG_POISON_ALLOC=poison)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "contention_soak.py"), "--procs", "8", "--rounds", "5", "--particles", "2000000",
                        "--steps", "12", "--shards", "--tenant", "--no-barrier", "--ladder", "native,torch", "--ladder-every-round", "--agree-after-first",
                        "--port", "29761" if poison == "0" else "29762", "--timeout", "400"], capture_output=True, text=True, timeout=600, env=env)
    rounds = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{"round"')]
    assert r.returncode == 0 and len(rounds) == 5, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    bad = [q for q in rounds if not q["equal"] or not all(q["shards_equal_own_whole"])]
    assert not bad, bad[:1]
    assert all(g == [0, True] for q in rounds for g in q["giveups"])  # (nobody degraded to the serial plan either: 2e6 particles take the multi-launch plan)
