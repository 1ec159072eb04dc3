#!/usr/bin/env python3
"""Per-message latency of nodes/pf_localizer_node: the reference's 300-step particle-filter demo scenario
(render_gif_particle_filter.rs:33-98; 150 particles, 5 landmarks) fed through the node's JSON-lines transport, five times per
configuration.  Round trip = feeder writes the range + odometry pair -> both output messages read back (Python feeder: JSON
encoding and decoding on its side included); step = what the node itself measures around try_step_state + publish (its log line).
    python tools/node_latency.py > profiles/r04_node_latency.json"""
import json
import os
import re
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import node_driver as D  # noqa: E402


def main():
    out = {"scenario": "render_gif_particle_filter.rs:33-98: 150 particles, 5 landmarks, 300 steps, x5", "unit": "microseconds", "rows": []}
    for transport in ("stdio", "unix"):
        for resident in ("20000", "0"):
            lat, step_mean, step_max = [], [], []
            for rep in range(5):
                with tempfile.TemporaryDirectory() as tmp:
                    node = D.NodeProcess(env={"PF_SEED": "42", "PF_RESIDENT_IDLE_US": resident, "PF_LOG_INTERVAL_S": "1000"}, transport=transport, tmpdir=tmp)
                    _, l, _ = D.run_scenario(node)
                    rc, err = node.close()
                    text = err + "\n".join(m.get("text", "") for m in node.logs)
                    m = re.search(r"step latency mean=([0-9.]+) us max=([0-9.]+) us", text)
                    assert rc == 0 and m, text[-500:]
                    lat.append(l[20:])
                    step_mean.append(float(m.group(1)))
                    step_max.append(float(m.group(2)))
            lat = np.concatenate(lat)
            out["rows"].append({"transport": transport, "resident_idle_us": int(resident), "round_trip_median": round(float(np.median(lat)), 1),
                                "round_trip_p99": round(float(np.percentile(lat, 99)), 1), "node_step_mean": round(float(np.mean(step_mean)), 1),
                                "node_step_max": round(float(np.max(step_max)), 1)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
