#!/bin/bash
# apply stage (500 x 4 planes) for several builds; IAP_HARNESS_SHIFT=x|y|yx makes those axes' shifts whole numbers
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/$v; fi
  echo "== $v"
  for m in ${MODES:-"" x y yx}; do
    echo -n "whole[$m] "; IAP_HARNESS_SHIFT=$m timeout -k 10 120 ./scripts/pmc_harness apply 500 4 || exit 1
  done
done
