for s in 5 20 10 20; do timeout -k 10 400 python bench.py --steps $s --warmup 2 --no-cpu-baseline --no-march --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('steps', d['steps'], round(d['value']), round(d['ms_per_step'],2), d['stages_ms_per_step'])"; done
