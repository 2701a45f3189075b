#!/bin/bash
# Round measurement run (GPU box): bench lines for every single-GPU configuration, the rocprofv3 kernel summary of
# the headline command, counter traffic for every stage.  usage: scripts/round_measure.sh <outdir>
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 500 python -u bench.py --stage-report > $out/c2_bench.json 2> $out/c2_bench.err || exit 1
echo c2 done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/c2_prof -o c2 --output-format csv -- python3 bench.py --no-traffic --no-cpu-baseline --no-verify > $out/c2_bench_under_rocprof.json 2> $out/c2_prof.err || exit 1
echo c2 rocprof done
for w in c1 c4fov c5 c3 blur smosigma; do
  timeout -k 10 600 python -u bench.py --workload $w --stage-report > $out/${w}_bench.json 2> $out/${w}_bench.err || { tail -5 $out/${w}_bench.err; exit 1; }
  echo $w done
done
timeout -k 10 300 python -u bench.py --workload c1 --reg-ref previous --no-kernel-timing --no-traffic > $out/c1_previous_bench.json 2> $out/c1_previous_bench.err || exit 1
timeout -k 10 600 python scripts/pmc_traffic.py --all --out $out/traffic.json > $out/traffic.log 2>&1 || { tail -5 $out/traffic.log; exit 1; }
echo traffic done
