cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/batch
for b in 64 16 8 4; do
  export IAP_HARNESS_BATCH=$b
  echo "== batch $b"
  timeout -k 10 120 ./scripts/pmc_harness estimate 128 || exit 1
  timeout -k 10 200 bash scripts/kstats.sh gpurun_out/batch/b$b estimate 128 || exit 1
done
