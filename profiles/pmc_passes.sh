#!/bin/bash
# usage: profiles/pmc_passes.sh <outdir> [batch] [traffic_batch] [extra bench.py flags, e.g. "--workload ext"]   (on the GPU box, through gpurun)
# One --kernel-trace --stats run, then one run per --pmc counter group (never combined with other trace domains);
# python profiles/summarize.py gpurun_out/<outdir> <tag> turns the CSVs into the committed summary.
# Every SQ pass also carries GRBM_GUI_ACTIVE (separate counter block), so each derived fraction is normalised with the
# shader cycles of ITS OWN pass.  The HBM-traffic passes (FETCH_SIZE, WRITE_SIZE) run at a smaller batch: at 65 536
# utterances they take many minutes under rocprofv3.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BATCH=${2:-65536}; TB=${3:-4096}; XF=${4:-}
B="python $R/bench.py $XF --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --batch $BATCH"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $R/bench.py $XF --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --batch $BATCH > $OUT/bench_under_rocprof.json 2>/dev/null
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  n=$(echo $c | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o r -- $B > /dev/null 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/traffic_${TB}_$c -o r -- python $R/bench.py $XF --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --batch $TB > /dev/null 2>&1
done
# the raw rocprofv3 output is far larger than what gpurun copies back (64 MiB): summarise here, keep only the summaries
cd $R
if [ -z "$XF" ]; then python profiles/summarize.py $OUT $1 $BATCH > /dev/null; else python profiles/summarize.py $OUT $1 $BATCH "$XF (B=$BATCH)" > /dev/null; fi
mkdir -p $R/gpurun_out/keep_$1
cp profiles/$1_rocprof_summary.csv $OUT/bench_under_rocprof.json $R/gpurun_out/keep_$1/
[ -z "$XF" ] && cp profiles/pmc_traffic.json profiles/pmc_valu.json $R/gpurun_out/keep_$1/
case "$XF" in *"--workload ext"*) cp profiles/pmc_valu_ext.json $R/gpurun_out/keep_$1/;; *"--templates 10"*) cp profiles/pmc_valu_k10.json $R/gpurun_out/keep_$1/;; *"--gain 2.4"*) cp profiles/pmc_valu_loud.json $R/gpurun_out/keep_$1/;; esac
rm -rf $OUT
ls $R/gpurun_out/keep_$1
