# Development aid (GPU box): C1 ('mean' and 'previous') bench lines for several builds.  usage: scripts/dev/c1_ab.sh <variant or "tree"> ...
for v in "$@"; do
  if [ "$v" = tree ]; then unset IAP_LIB; else export IAP_LIB=$GRAFT_REPO_ROOT/build/$v/libiap_hip.so; fi
  echo "== $v"
  for r in mean previous; do
    timeout -k 10 200 python bench.py --workload c1 --reg-ref $r --no-traffic --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$r', round(d['ms_per_step'],4), d.get('stage_ms_per_step'), d['verified'].get('max_shift_error_px'))" || exit 1
  done
done
