#!/bin/bash
# Brent's state in scalar registers (fit_kernel.hpp: MET2_UNI_MASK) -- the workloads whose kernels the mask touches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-uni}
run() {
    timeout -k 5 600 python3 $R/bench.py --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', '%-70s' % d['config']['workload'][:70], '%9.0f voxels/s' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'kernel %.2f' % r['kernel_ms'], 'second %.2f' % r['second_pass_ms'], 'spill', r.get('spill_voxels'))"
}
run --config 1 --steps 3 --warmup 1
run --config 3 --dims 128,128,64 --steps 2 --warmup 1
run --config 4 --dims 64,64,32 --nte 32 --nt2 60 --fa single --steps 2 --warmup 1
run --dims 32,32,32 --method X2 --penalty L2 --nte 48 --nt2 120 --steps 3 --warmup 1
run --dims 32,32,32 --method BayesReg --penalty InvT2 --nte 48 --nt2 120 --steps 3 --warmup 1
run --dims 64,64,32 --method GCV --penalty L2 --nte 48 --nt2 120 --steps 2 --warmup 1
