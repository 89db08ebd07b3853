#!/bin/bash
# HBM traffic of the dominant fit kernel of one bench workload, two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), bytes per voxel:
#   bash scripts/dev/traffic.sh LABEL [bench args, default --config 1]
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-x}; shift
ARGS=${@:---config 1}
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/tr_$c -o t --output-format csv -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > /tmp/tr_$c.log 2>&1 || { echo "pass $c failed"; tail -3 /tmp/tr_$c.log; }
done
python3 - "$L" <<'PY'
import csv, glob, sys
tot = {}
vox = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/tr_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "fit_kernel" in r["Kernel_Name"] and ", false>" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by[r["Dispatch_Id"]] = by.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    tot[c] = max(by.values())
import json
line = [l for l in open("/tmp/tr_WRITE_SIZE.log") if l.startswith("{")][-1]
d = json.loads(line); vox = d["config"]["fitted_voxels_per_gpu"]
b = 1024.0 * (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"])
print(sys.argv[1], "fetch KB %.0f write KB %.0f -> %.0f B/voxel (%.2fx of %d algorithmic), kernel %.2f ms" % (tot["FETCH_SIZE"], tot["WRITE_SIZE"], b / vox, b / vox / d["roofline"]["bytes_per_voxel"], d["roofline"]["bytes_per_voxel"], d["roofline"]["kernel_ms"]))
PY
