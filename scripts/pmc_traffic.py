"""HBM traffic per kernel launch from rocprofv3 counter passes over the in-tree libiap_hip.so.

`rocprofv3 --pmc` cannot profile python/torch on this image, so every pass runs scripts/pmc_harness (a native
program that calls the library's C-ABI on the benchmark's shapes).  One run per counter: FETCH_SIZE and WRITE_SIZE
do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Both are kilobytes at the L2's memory side; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled (same guide, HBM section).

    python scripts/pmc_traffic.py apply 500 4                 # one stage, prints JSON
    python scripts/pmc_traffic.py --all --out profiles/traffic.json

bench.py imports `measure()` for the `roofline.traffic` field of its line.
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "scripts", "pmc_harness")


_KERNEL = re.compile(r"iap::(\w+)|_ZN3iap\d+([A-Za-z0-9_]+?kernel)")


def _per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            m = _KERNEL.search(r["Kernel_Name"])      # demangled "iap::name<...>" or mangled "_ZN3iap<len>name..."
            if r["Counter_Name"] != counter or not m:
                continue
            a = acc[m.group(1) or m.group(2)]
            a[0] += float(r["Counter_Value"]) * 1024.0
            a[1] += 1
    return acc


def _pass(counter, harness_args, workdir, timeout):
    out = os.path.join(workdir, counter)
    cmd = ["rocprofv3", "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--", HARNESS] + [str(a) for a in harness_args]
    env = dict(os.environ, IAP_HARNESS_REPS="2")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} -> {r.returncode}: {r.stderr[-400:]}")
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise RuntimeError(f"no counter_collection.csv under {out}")
    timing = None
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            timing = json.loads(line)
    return _per_kernel(files[0], counter), timing


def measure(harness_args, timeout=120, keep=None):
    """{kernel: {hbm_bytes_per_launch, fetch_bytes_per_launch, write_bytes_per_launch, launches_sampled}} for one
    harness invocation, e.g. measure(["apply", 500, 4]).  Raises when rocprofv3 or the harness is unavailable."""
    if not os.access(HARNESS, os.X_OK):
        raise RuntimeError(f"{HARNESS} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    work = keep or tempfile.mkdtemp(prefix="iap_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    os.makedirs(work, exist_ok=True)
    try:
        fetch, timing = _pass("FETCH_SIZE", harness_args, work, timeout)
        write, _ = _pass("WRITE_SIZE", harness_args, work, timeout)
    finally:
        if keep is None:
            shutil.rmtree(work, ignore_errors=True)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        fb = 2.0 * f / nf if nf else None
        wb = w / nw if nw else None
        calls = 3.0       # the harness runs its stage once to warm up and IAP_HARNESS_REPS = 2 times more
        out[k] = {"hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0), "fetch_bytes_per_launch": fb,
                  "write_bytes_per_launch": wb, "launches_sampled": max(nf, nw),
                  # kernels launched several times per call with unequal sizes (the estimate's: one small launch
                  # for the reference, one per 64-frame group): the sum over a call is the meaningful figure
                  "hbm_bytes_per_harness_call": (2.0 * f + w) / calls,
                  "config": "scripts/pmc_harness " + " ".join(str(a) for a in harness_args) + " (C-ABI of libiap_hip.so)",
                  "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, rocprofv3 --pmc, one pass per counter"}
        if timing:
            out[k]["harness_ms_per_call_under_profiler"] = timing.get("ms_per_call")
    return out


ALL = (["apply", 500, 4], ["bg", 2000], ["estimate", 128], ["measure", 500, 4], ["measure", 64, 1, 2048, 2048, "dense"],
       ["blur", 500], ["smosigma", 200])


if __name__ == "__main__":
    args = sys.argv[1:]
    outp = None
    if "--out" in args:
        i = args.index("--out")
        outp = args[i + 1]
        del args[i:i + 2]
    runs = ALL if "--all" in args else [args]
    res = {}
    for ha in runs:
        got = measure(ha)
        for k, v in got.items():
            key = k if k not in res else f"{k} ({' '.join(str(a) for a in ha)})"
            res[key] = v
            print(f"{key:44s} fetch {v['fetch_bytes_per_launch'] or 0:16.0f}  write {v['write_bytes_per_launch'] or 0:16.0f}  "
                  f"launches {v['launches_sampled']}", flush=True)
    if outp:
        with open(outp, "w") as fh:
            json.dump(res, fh, indent=1)
    else:
        print(json.dumps(res))
