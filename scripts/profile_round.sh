#!/bin/bash
# Evidence for DESIGN.md section 6, run on the GPU box through gpurun:
#   bash scripts/profile_round.sh TAG
# writes gpurun_out/TAG_bench.json (bench line incl. CPU baseline), TAG_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
# TAG_pmc_fetch.csv / TAG_pmc_write.csv (separate --pmc passes).  Copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 5 400 python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1 || exit 2
cp $OUT/${TAG}_stats/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pf -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_pf.log 2>&1 || exit 3
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pw -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_pw.log 2>&1 || exit 4
python3 - <<PY
import csv
for tag, f in (("fetch", "$OUT/${TAG}_pf/f_counter_collection.csv"), ("write", "$OUT/${TAG}_pw/w_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "fit_kernel" in r["Kernel_Name"] or "nesma" in r["Kernel_Name"]]
    with open("$OUT/${TAG}_pmc_%s.csv" % tag, "w", newline="") as o:
        w = csv.writer(o); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id"])
        for r in rows: w.writerow([r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"], r["Dispatch_Id"]])
    print(tag, [(r["Kernel_Name"][:40], r["Counter_Value"]) for r in rows])
PY
head -c 600 $OUT/${TAG}_bench.json; echo
head -5 $OUT/${TAG}_kernel_stats.csv
