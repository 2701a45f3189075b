#!/bin/bash
# Development aid: per-kernel durations of one harness run.  usage: scripts/kstats.sh <outdir> <harness args...>
out=$1; shift
mkdir -p $out
export IAP_HARNESS_REPS=2
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $out -o k --output-format csv -- ./scripts/pmc_harness "$@" > $out/k.log 2>&1 || { tail -5 $out/k.log; exit 1; }
python3 - $out <<'PY'
import csv,sys,glob,re
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'iap' in r['Name']:
            nm = (re.search('iap(?:::|[0-9]+)([A-Za-z_][A-Za-z_0-9]*?kernel)', r['Name']) or re.search('(iap.{0,40})', r['Name'])).group(1)
            print("%-28s calls %5s avg %10.1f us total %9.3f ms" % (nm, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
