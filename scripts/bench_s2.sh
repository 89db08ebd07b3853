#!/bin/bash
# The three nTE=48, nT2=120 lines of bench_other_methods.sh on their own (A/B runs of solver changes at two bins per lane)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for a in "--dims 32,32,32 --method X2 --penalty L2" "--dims 32,32,32 --method BayesReg --penalty InvT2" "--dims 64,64,32 --method GCV --penalty L2"; do
    timeout -k 5 300 python3 $R/bench.py --no-cpu-baseline $a --nte 48 --nt2 120 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), d['roofline']['kernel_ms'])"
done
