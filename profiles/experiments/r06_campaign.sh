#!/bin/bash
# The round's measurement campaign on ONE box (through gpurun): rocprofv3 stats + PMC passes for every benchmark shape, the
# default bench line, both sustained runs, the 8-rank one-device dry runs.  Outputs under gpurun_out/ (copied to profiles/ by hand).
#   gpurun --timeout 3000 -- 'bash profiles/experiments/r06_campaign.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
bash profiles/pmc_passes.sh r06_v2 > gpurun_out/camp_v2.log 2>&1
bash profiles/pmc_passes.sh r06_loud 65536 4096 "--gain 2.4" > gpurun_out/camp_loud.log 2>&1
bash profiles/pmc_passes.sh r06_ext 65536 4096 "--workload ext" > gpurun_out/camp_ext.log 2>&1
bash profiles/pmc_passes.sh r06_k10 4096 4096 "--batch 4096 --templates 10" > gpurun_out/camp_k10.log 2>&1
bash profiles/pmc_dp.sh r06 > gpurun_out/camp_dp.log 2>&1
cd $R
python bench.py --steps 20 --warmup 3 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
bash profiles/experiments/sustained.sh r06_sustained 2000 > gpurun_out/r06_sustained.log 2>&1
bash profiles/experiments/sustained.sh r06_sustained_loud 1000 "--gain 2.4" > gpurun_out/r06_sustained_loud.log 2>&1
SR_BENCH_BACKEND=gloo SR_BENCH_DEVICE=0 python bench.py --gpus 8 --batch 8192 --steps 3 --warmup 1 > gpurun_out/r06_n8_dryrun_bench.json 2> gpurun_out/n8.err
SR_BENCH_BACKEND=gloo SR_BENCH_DEVICE=0 python bench.py --gpus 8 --batch 8192 --steps 3 --warmup 1 --exchange results > gpurun_out/r06_n8_dryrun_results_bench.json 2>> gpurun_out/n8.err
ls -la gpurun_out | tail -30
