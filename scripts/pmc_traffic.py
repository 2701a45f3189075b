"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs into profiles/traffic.json.

Usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are kilobytes at the L2's memory side; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section).  Values are per launch.
"""
import csv, json, re, sys, collections

def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "iap::" not in r["Kernel_Name"]:
            continue
        m = re.search(r"iap::(\w+)", r["Kernel_Name"])
        a = acc[m.group(1)]
        a[0] += float(r["Counter_Value"]) * 1024.0
        a[1] += 1
    return acc

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
    fb = 2.0 * f / nf if nf else None
    wb = w / nw if nw else None
    out[k] = {"hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0), "fetch_bytes_per_launch": fb,
              "write_bytes_per_launch": wb, "launches_sampled": max(nf, nw),
              "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, rocprofv3 --pmc, separate passes"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:28s} fetch {v['fetch_bytes_per_launch'] or 0:14.0f}  write {v['write_bytes_per_launch'] or 0:14.0f}  launches {v['launches_sampled']}")
