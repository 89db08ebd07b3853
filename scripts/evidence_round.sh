#!/bin/bash
# Round evidence, run on the GPU box:  [STAGE=pmc1|pmc2|pmc3|bench|parity] bash scripts/evidence_round.sh TAG
#   gpurun_out/TAG_bench_configs.jsonl         the five BASELINE configs at full size (bench lines with roofline, cpu_baseline, parity, end_to_end)
#   gpurun_out/TAG_<key>_{pmc,kernel_stats}.csv + pmc_counters.json   rocprofv3 kernel-trace + PMC passes per workload (full size for every config)
#   gpurun_out/parity_TAG.json, kat_TAG.json   parity rates on the reference fixtures (4 096 / 512 voxels per method, 65 536 for X2/L2), statistical known-answer run
# The whole set takes ~30 GPU-minutes; gpurun calls are limited to 20, so the stages are run as separate calls (pmc_counters.json accumulates:
# copy gpurun_out/pmc_counters.json into profiles/ between calls).
TAG=${1:-rXX}
STAGE=${STAGE:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
pmc() { key=$1; shift; python3 scripts/collect_pmc.py --tag $TAG --key $key "$@" > gpurun_out/${TAG}_pmc_$key.log 2>&1 || echo "pmc $key failed"; }
if [ $STAGE = all ] || [ $STAGE = pmc1 ]; then
  pmc config1 -- --config 1
  pmc config0 -- --config 0
  pmc config2 -- --config 2
  pmc config3 -- --config 3
fi
if [ $STAGE = all ] || [ $STAGE = pmc2 ]; then
  pmc config4 --timeout 1100 -- --config 4
fi
if [ $STAGE = all ] || [ $STAGE = pmc3 ]; then
  pmc config4_131k_fa --kernel fa_kernel --sum -- --config 4 --dims 64,64,32
  pmc config4_131k_fa_gemm --kernel fa_project_kernel --sum -- --config 4 --dims 64,64,32
  pmc gcv_s1_131k -- --config 4 --dims 64,64,32 --nte 32 --nt2 60 --fa single
  pmc bayes_s2_32k -- --config 3 --dims 32,32,32 --nte 48 --nt2 120
  pmc x2_s2_32k -- --config 1 --dims 32,32,32 --nte 48 --nt2 120
  pmc tv_128x128x64x32 --kernel tv_iter_kernel --sha-files met2_tv.hip -- --workload tv
fi
if [ $STAGE = all ] || [ $STAGE = bench ]; then
  bash scripts/bench_configs.sh $TAG
  bash scripts/bench_other_methods.sh $TAG
  python3 scripts/dev/driver_probe.py > gpurun_out/${TAG}_driver_probe.jsonl 2> gpurun_out/${TAG}_driver_probe.err
  python3 bench.py --workload tv > gpurun_out/${TAG}_tv_bench.json 2> gpurun_out/${TAG}_tv_bench.err
  python3 scripts/dev/pipeline_probe.py > gpurun_out/${TAG}_pipeline_timing.jsonl 2> gpurun_out/${TAG}_pipeline_timing.err
fi
if [ $STAGE = all ] || [ $STAGE = parity ]; then
  python3 tests/tools/parity_report.py --out gpurun_out/parity_${TAG}.json > gpurun_out/parity_${TAG}.log 2>&1
  python3 tests/tools/kat_report.py -n 20000 --out gpurun_out/kat_${TAG}.json > gpurun_out/kat_${TAG}.log 2>&1
fi
ls gpurun_out | grep "^${TAG}_\|_${TAG}" | head -60
