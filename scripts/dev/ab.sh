#!/bin/bash
# A/B of build-time switches (run on the GPU box):  [ONLY=<method number>] [BENCH="--config 3"] scripts/dev/ab.sh "-DFOO=0" "-DFOO=1" ...
# Each argument is a set of extra defines; the single-method library (-DMET2_ONLY, default 2 = X2) is rebuilt for each and bench.py's
# kernel time and parity block printed.  The tree's library is left at the LAST build: rebuild before anything else.
cd "$(dirname "$0")/.."
ONLY=${ONLY:-2}
for defs in "$@"; do
  MET2_BUILD_DEFINES="-DMET2_ONLY=$ONLY $defs" python3 -c "import importlib; importlib.import_module('multicomponent-t2-toolbox_amd._build').build(force=True)" || exit 1
  MET2_BUILD_DEFINES="-DMET2_ONLY=$ONLY $defs" python3 bench.py --steps 3 --warmup 1 --no-end-to-end --cpu-seconds 6 --parity-sample 40000 $BENCH | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('DEFS [$defs] fa_ms %.2f kernel_ms %.2f pass2 %.2f value %.0f parity over1e-5 %d/%d max %.2e maxdMWF %.2e' % (d['roofline'].get('fa_kernel_ms', 0.0), d['roofline']['kernel_ms'], d['roofline']['second_pass_ms'], d['value'], d['parity']['n_over_1e-5'], d['parity']['sample'], d['parity']['max_rel_fsol'], d['parity']['max_abs_MWF']))"
done
