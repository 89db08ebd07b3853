#!/bin/bash
# Round-5 A/B lines (one GPU): kernel ms, second-pass ms and spill-over voxels of the workloads the single-launch fit is judged on.
#   bash scripts/dev/ab5.sh [label]      (MET2_TWO_PASS=1 in the environment: the two-launch capacity ladder of rounds 1-4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-ab}
run() {
    timeout -k 5 600 python3 $R/bench.py --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', '%-95s' % d['config']['workload'][:95], '%10.0f voxels/s' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'kernel %.2f' % r['kernel_ms'], 'second %.2f' % r['second_pass_ms'], 'spill', r.get('spill_voxels'), 'fa', r.get('fa_kernel_ms'))"
}
run --config 1 --steps 3 --warmup 1
run --config 2 --dims 128,128,64 --steps 2 --warmup 1
run --dims 32,32,32 --method X2 --penalty L2 --nte 48 --nt2 120 --steps 3 --warmup 1
run --dims 32,32,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 3 --warmup 1
run --dims 32,32,32 --method BayesReg --penalty InvT2 --nte 48 --nt2 120 --steps 3 --warmup 1
run --dims 64,64,32 --method GCV --penalty L2 --nte 48 --nt2 120 --steps 2 --warmup 1
run --config 4 --dims 64,64,32 --steps 2 --warmup 1
