#!/bin/bash
# Marginal cost of the solver's phases on configs[1] (run on the GPU box): rebuilds the X2 kernel with one idempotent phase
# executed twice (-DMET2_DOUBLE=k) and prints the kernel time of each build.  The tree's library is restored at the end.
set -e
cd "$(dirname "$0")/.."
for k in 0 1 2 3 4; do
  MET2_BUILD_DEFINES="-DMET2_ONLY=2 -DMET2_DOUBLE=$k" python3 -c "import importlib; importlib.import_module('multicomponent-t2-toolbox_amd._build').build(force=True)"
  MET2_BUILD_DEFINES="-DMET2_ONLY=2 -DMET2_DOUBLE=$k" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('MET2_DOUBLE=$k kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'])"
done
