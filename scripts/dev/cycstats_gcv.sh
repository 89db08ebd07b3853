#!/bin/bash
# Wave-cycle shares of the phases of fit_kernel<GCV_LR, 2> (48x120, 131 072 voxels): a -DMET2_CYCSTATS build (development counters), MET2_DEBUG prints them.
# Builds into the tree on the GPU box and restores the default library afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
P=multicomponent-t2-toolbox_amd
cp $P/libmet2_hip.so /tmp/libmet2_keep.so; cp $P/libmet2_hip.so.flags /tmp/libmet2_keep.flags 2>/dev/null
MET2_BUILD_DEFINES="-DMET2_ONLY=4 -DMET2_CYCSTATS" python3 -c "import importlib; importlib.import_module('multicomponent-t2-toolbox_amd._build').build(force=True)" > /tmp/cyc_build.log 2>&1 || { tail -20 /tmp/cyc_build.log; exit 1; }
MET2_BUILD_DEFINES="-DMET2_ONLY=4 -DMET2_CYCSTATS" MET2_DEBUG=1 timeout -k 5 600 python3 bench.py --dims 64,64,32 --method GCV --penalty L2 --nte 48 --nt2 120 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end 2>&1 | grep "\[met2\] \(gcv\|calls\|wave cycles\)" | tail -6
cp /tmp/libmet2_keep.so $P/libmet2_hip.so; cp /tmp/libmet2_keep.flags $P/libmet2_hip.so.flags 2>/dev/null
