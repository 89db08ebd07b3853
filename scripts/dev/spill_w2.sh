#!/bin/bash
# The spill-over kernel's duration against the number of waves per workgroup its LDS is carved for (MET2_SPILL_W2; 0 = the kernel's own choice).
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
    L=$1; shift
    timeout -k 5 600 python3 $R/bench.py --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', '%-70s' % d['config']['workload'][:70], 'ms/step %.2f' % d['ms_per_step'], 'kernel %.2f' % r['kernel_ms'], 'second %.2f' % r['second_pass_ms'], 'spill', r.get('spill_voxels'))"
}
for w in 0 2 3 4 5 6 8; do
export MET2_SPILL_W2=$w
run w2=$w --dims 32,32,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 3 --warmup 1
run w2=$w --dims 64,64,32 --method L_curve --penalty L1 --nte 48 --nt2 120 --steps 2 --warmup 1
run w2=$w --dims 32,32,32 --method X2 --penalty L2 --nte 48 --nt2 120 --steps 3 --warmup 1
run w2=$w --dims 64,64,32 --method GCV --penalty L2 --nte 48 --nt2 120 --steps 2 --warmup 1
done
