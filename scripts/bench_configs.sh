#!/bin/bash
# All five BASELINE configs at their stated sizes on ONE GPU, full bench lines (roofline + cpu_baseline + parity):
#   bash scripts/bench_configs.sh TAG      -> gpurun_out/TAG_bench_configs.jsonl   (copy into profiles/ to have it judged)
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
: > $OUT/${TAG}_bench_configs.jsonl
for c in 0 1 2 3; do
  timeout -k 10 600 python3 $R/bench.py --config $c --steps 3 --warmup 1 >> $OUT/${TAG}_bench_configs.jsonl 2>> $OUT/${TAG}_bench_configs.err || exit 1
done
timeout -k 10 900 python3 $R/bench.py --config 4 --steps 1 --warmup 0 >> $OUT/${TAG}_bench_configs.jsonl 2>> $OUT/${TAG}_bench_configs.err || exit 2
python3 - <<PY
import json
for l in open("$OUT/${TAG}_bench_configs.jsonl"):
    d = json.loads(l)
    print("%-110s %12.0f voxels/s  kernel %.1f ms  cpu %s  parity %s" % (d["config"]["workload"][:110], d["value"], d["roofline"]["kernel_ms"],
          d.get("cpu_baseline", {}).get("value"), d.get("parity", {}).get("frac_over_1e-5")))
PY
