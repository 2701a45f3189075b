#!/bin/bash
# C2 bench line for several builds of the library, alternating (GPU box).  usage: scripts/dev/bench_ab.sh <variant dir or "tree"> ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset IAP_LIB; else export IAP_LIB=$GRAFT_REPO_ROOT/build/$v/libiap_hip.so; fi
  timeout -k 10 200 python -u bench.py --no-traffic --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'ms/step %.2f' % d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()}, round(d['roofline']['frac'], 3))
" || exit 1
done
