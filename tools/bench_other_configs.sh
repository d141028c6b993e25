for c in C1 C2 C3 C5; do python bench.py --config $c --steps 3 --warmup 1 --cpu-seconds 8 --no-alt-builder 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$c', d['config']['workload'], '|', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per step | parity', d['parity']['linear_radiance_bit_exact'], d['parity']['pixels'], 'px', d['parity']['spp'], 'spp | cpu', d['cpu_baseline']['value'], 'Mrays/s | builder', d['config']['bvh_builder'], d['config']['bvh_build_s'], 's')"; done
