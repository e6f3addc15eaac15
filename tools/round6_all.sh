#!/bin/bash
# Runs on the GPU box: everything profiles/r6_* is made of.
repo=${GRAFT_REPO_ROOT:-$(pwd)}
cd $repo
bash tools/profile_round.sh r6 > gpurun_out/r6_profile_round.log 2>&1
bash tools/pmc_round.sh r6 > gpurun_out/r6_pmc_round.log 2>&1
bash tools/split_pmc.sh r6 > gpurun_out/r6_split_pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out r6 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out r6 transformer > /dev/null 2>&1
python3 tools/pmc_utilisation_json.py gpurun_out r6 > /dev/null 2>&1
python3 tools/ops_cost.py > gpurun_out/r6_ops_cost.json 2> gpurun_out/r6_ops_cost.err
ls gpurun_out | grep r6_ | head -60
cat gpurun_out/r6_bench.json
tail -3 gpurun_out/r6_bench.err
