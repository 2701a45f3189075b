#!/bin/bash
# bg levels stage (SMO march + median) for several builds (GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/$v; fi
  echo "== $v"
  for k in 1 2; do
  timeout -k 10 120 ./scripts/pmc_harness bg 2000 || exit 1
  done
done
