#!/bin/bash
# Round evidence, run on the GPU box:  bash scripts/evidence_round.sh TAG
#   gpurun_out/TAG_bench_configs.jsonl         the five BASELINE configs at full size (bench lines with roofline, cpu_baseline, parity)
#   gpurun_out/TAG_<key>_{pmc,kernel_stats}.csv + pmc_counters.json   rocprofv3 kernel-trace + PMC passes per workload
#   gpurun_out/parity_TAG.json, kat_TAG.json   parity rates on the 4 096-voxel reference fixtures, statistical known-answer run
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
rm -f gpurun_out/pmc_counters.json
python3 scripts/collect_pmc.py --tag $TAG --key config1 -- --config 1 > gpurun_out/${TAG}_pmc_config1.log 2>&1 || echo "pmc config1 failed"
python3 scripts/collect_pmc.py --tag $TAG --key config0 -- --config 0 > gpurun_out/${TAG}_pmc_config0.log 2>&1 || echo "pmc config0 failed"
python3 scripts/collect_pmc.py --tag $TAG --key config2 -- --config 2 > gpurun_out/${TAG}_pmc_config2.log 2>&1 || echo "pmc config2 failed"
python3 scripts/collect_pmc.py --tag $TAG --key config3 -- --config 3 > gpurun_out/${TAG}_pmc_config3.log 2>&1 || echo "pmc config3 failed"
python3 scripts/collect_pmc.py --tag $TAG --key config4_131k -- --config 4 --dims 64,64,32 > gpurun_out/${TAG}_pmc_config4.log 2>&1 || echo "pmc config4 failed"
python3 scripts/collect_pmc.py --tag $TAG --key config4_131k_fa --kernel fa_kernel -- --config 4 --dims 64,64,32 > gpurun_out/${TAG}_pmc_config4fa.log 2>&1 || echo "pmc config4 fa failed"
python3 scripts/collect_pmc.py --tag $TAG --key gcv_s1_131k -- --config 4 --dims 64,64,32 --nte 32 --nt2 60 --fa single > gpurun_out/${TAG}_pmc_gcv_s1.log 2>&1 || echo "pmc gcv s1 failed"
python3 scripts/collect_pmc.py --tag $TAG --key bayes_s2_32k -- --config 3 --dims 32,32,32 --nte 48 --nt2 120 > gpurun_out/${TAG}_pmc_bayes_s2.log 2>&1 || echo "pmc bayes s2 failed"
bash scripts/bench_configs.sh $TAG
python3 tests/tools/parity_report.py --out gpurun_out/parity_${TAG}.json > gpurun_out/parity_${TAG}.log 2>&1
python3 tests/tools/kat_report.py -n 20000 --out gpurun_out/kat_${TAG}.json > gpurun_out/kat_${TAG}.log 2>&1
ls gpurun_out | grep "^${TAG}_" | head -50
