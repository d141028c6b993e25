#!/bin/bash
# C4 (bench.py, 256 spp) over the leaf-pass policy: leaf hold x postpone percentage x leaf quorum.  Usage (GPU box): tools/c4_policy_sweep.sh > gpurun_out/c4_policy.txt
for h in 4 3 2; do for p in 25 40 55; do for q in 1 4; do
  printf "hold %s pct %s quorum %s: " $h $p $q
  HRT_LEAF_HOLD=$h HRT_POSTPONE_PCT=$p HRT_LEAF_QUORUM=$q timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-builder 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done; done; done
