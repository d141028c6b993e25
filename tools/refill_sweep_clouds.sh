for v in 8 12 16 20; do echo "== HRT_REFILL_THRESHOLD=$v"; HRT_REFILL_THRESHOLD=$v python3 tools/two_level_bench.py --render-only --spp 4 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ', d['scene'], d['structure'], d['spp4']['Mrays_per_s'])"; done
