#!/bin/bash
# Development aid: build libiap_hip.so with extra -D flags into build/<name>/ (run the harness against it with
# LD_LIBRARY_PATH=build/<name>).   usage: scripts/build_variant.sh <name> [-DFOO=1 ...]
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
out=$root/build/$name
mkdir -p $out
cd $root/imageanalysis_pipeline_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c $f -o $out/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $out/*.o -o $out/libiap_hip.so
rm -f $out/*.o
echo built $out/libiap_hip.so
