#!/bin/bash
# estimate stage against the registration launch group size (GPU box)
for b in ${BATCHES:-64 72 84 100 125 167}; do
  timeout -k 10 120 python -u bench.py --batch $b --no-traffic --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch', $b, 'ms/step %.2f' % d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
" || exit 1
done
