#!/bin/bash
# A/B inside ONE build: who sends a shard's weight maximum to the ranks' mailboxes -- the step kernel's last workgroup
# (RR_P2P_WMAX_EARLY=1, the default) or the plan kernel's first (=0) -- on the sharded world-1 p2p step, alternating, same box
cd /root/repo
for i in 1 2 3 4; do
  for early in 0 1; do
    RR_P2P_WMAX_EARLY=$early python bench.py --gpus 1 --steps 200 --warmup 20 --force-sharded --transport p2p-only --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('RR_P2P_WMAX_EARLY=$early', round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done
