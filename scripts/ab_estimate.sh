#!/bin/bash
# Development aid (GPU box): per-kernel durations of the estimate stage for several builds of the library.
# usage: scripts/ab_estimate.sh <outdir> <frames> <variant dir or "tree"> ...
out=$1; frames=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/$v; fi
  echo "== $v"
  timeout -k 10 120 ./scripts/pmc_harness estimate $frames || exit 1
  timeout -k 10 200 bash scripts/kstats.sh $out/$v estimate $frames || exit 1
done
