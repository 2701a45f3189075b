#!/bin/bash
# measure stage, sparse (C2) and dense (C5) labels, for several builds (GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/$v; fi
  echo "== $v"
  for k in 1 2; do
  timeout -k 10 120 ./scripts/pmc_harness measure 500 4 || exit 1
  timeout -k 10 120 ./scripts/pmc_harness measure 64 1 2048 2048 dense || exit 1
  timeout -k 10 120 ./scripts/pmc_harness measure 300 3 || exit 1
  done
done
