#!/bin/bash
# Development aid (GPU box): estimate stage at several plane sizes, 8 against 16 points per thread in the row kernels
# (build/e8 = scripts/build_variant.sh e8 -DC2R_E=8 -DR2C_E=8)
cd $GRAFT_REPO_ROOT
for v in e8 tree e8 tree; do
  if [ "$v" = tree ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/$v; fi
  echo "== $v"
  timeout -k 10 60 ./scripts/pmc_harness estimate 512 1024 1024 || exit 1
  timeout -k 10 60 ./scripts/pmc_harness estimate 64 4096 4096 || exit 1
  timeout -k 10 60 ./scripts/pmc_harness estimate 256 2048 1024 || exit 1
done
