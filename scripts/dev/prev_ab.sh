# Development aid (GPU box): per-kernel durations of the serial 'previous' chain on a C1 stack for several builds.
# usage: scripts/dev/prev_ab.sh <outdir> <variant dir or "tree"> ...
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset IAP_LIB; else export IAP_LIB=$GRAFT_REPO_ROOT/build/$v/libiap_hip.so; fi
  echo "== $v"
  mkdir -p $out/$v
  timeout -k 10 200 python bench.py --workload c1 --reg-ref previous --no-kernel-timing --no-traffic --no-cpu-baseline --no-verify 2>/dev/null | cut -c1-200 || exit 1
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $out/$v -o k --output-format csv -- python3 bench.py --workload c1 --reg-ref previous --no-kernel-timing --no-traffic --no-cpu-baseline --no-verify > $out/$v/k.log 2>&1 || { tail -5 $out/$v/k.log; exit 1; }
  python3 - $out/$v <<'PY'
import csv,sys,glob,re
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'iap' in r['Name']:
            nm = (re.search('iap(?:::|[0-9]+)([A-Za-z_][A-Za-z_0-9]*?kernel)', r['Name']) or re.search('(iap.{0,40})', r['Name'])).group(1)
            print("%-28s calls %5s avg %10.1f us total %9.3f ms" % (nm, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
