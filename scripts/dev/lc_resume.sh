#!/bin/bash
# L-curve at two bins per lane: the spill-over kernel going on from the saved sweep state (default) against starting over (MET2_LC_RESTART=1).
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
    L=$1; shift
    timeout -k 5 600 python3 $R/bench.py --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', '%-80s' % d['config']['workload'][:80], '%10.0f voxels/s' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'kernel %.2f' % r['kernel_ms'], 'second %.2f' % r['second_pass_ms'], 'spill', r.get('spill_voxels'))"
}
run resume --dims 32,32,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 3 --warmup 1
MET2_LC_RESTART=1 run restart --dims 32,32,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 3 --warmup 1
run resume --dims 64,64,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 3 --warmup 1
MET2_LC_RESTART=1 run restart --dims 64,64,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 3 --warmup 1
run resume --config 2 --dims 128,128,64 --steps 2 --warmup 1
