#!/bin/bash
# Development aid: SQ activity counters of one harness stage.  usage: scripts/pmc_sq.sh <outdir> <harness args...>
out=$1; shift
mkdir -p $out
export IAP_HARNESS_REPS=1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $out -o a --output-format csv -- ./scripts/pmc_harness "$@" > $out/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $out -o b --output-format csv -- ./scripts/pmc_harness "$@" > $out/b.log 2>&1 || exit 1
python3 - $out <<'PY'
import csv,sys,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'iap(?:::|[0-9]+)([A-Za-z_][A-Za-z_0-9]*?kernel)',r['Kernel_Name'])
        if not m: continue
        k=m.group(1)
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
        acc[k]['_dur_'+r['Counter_Name']]+=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        if c.startswith('_'): continue
        n=cnt[k][c]; print(f"   {c:24s} {acc[k][c]/n:16.0f} per launch   ({n} launches, {acc[k]['_dur_'+c]/n/1e3:.1f} us)")
PY
