#!/bin/bash
# Secondary bench lines (other lambda-selection methods and the S2 shape); one JSON line each into gpurun_out/TAG_other.jsonl
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_other.jsonl
: > $O
run() { timeout -k 5 300 python3 $R/bench.py --no-cpu-baseline "$@" >> $O 2>> $R/gpurun_out/${TAG}_other.err || echo "{\"failed\": \"$*\"}" >> $O; }
run --dims 64,64,64 --method L_curve --penalty L1
run --dims 64,64,64 --method BayesReg --penalty InvT2
run --dims 64,64,32 --method GCV --penalty L2
run --dims 64,64,64 --method NNLS --penalty I
run --dims 64,64,64 --method T2SPARC --penalty InvT2
run --dims 32,32,32 --method X2 --penalty L2 --nte 48 --nt2 120
run --dims 32,32,32 --method BayesReg --penalty InvT2 --nte 48 --nt2 120
run --dims 64,64,32 --method GCV --penalty L2 --nte 48 --nt2 120
run --dims 64,64,64 --method X2 --penalty L2 --fa brute-force
python3 - <<PY
import json
for l in open("$O"):
    d = json.loads(l)
    print(d.get("config", d).get("workload", d) if "config" in d else d, "->", d.get("value"))
PY
