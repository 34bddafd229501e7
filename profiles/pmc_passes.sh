#!/bin/bash
# usage: profiles/pmc_passes.sh <outdir> <batch>   (on the GPU box, through gpurun)
# One --kernel-trace --stats run, then one run per --pmc counter group (never combined with other trace domains);
# python profiles/summarize.py gpurun_out/<outdir> <tag> turns the CSVs into the committed summary.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch $2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --batch $2 > $OUT/bench_under_rocprof.json 2>/dev/null
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE"; do   # FETCH_SIZE WRITE_SIZE (HBM traffic, profiles/pmc_traffic.json) is a separate pass that takes many minutes under
         # rocprofv3 with this workload (a 15-minute attempt at B = 16 384 did not finish): the committed figure is the r01_v4 one
  n=$(echo $c | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o r -- $B > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(dict)
for d in sorted(glob.glob('$OUT/pmc_*')):
    fs = glob.glob(d+'/*counter_collection.csv')
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if k.startswith('sr::'): acc[k.split('(')[0]][r['Counter_Name']] = float(r['Counter_Value'])
for k in acc:
    print(k, ' '.join(f'{c}={v:.4g}' for c,v in sorted(acc[k].items())))
PY
