#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sp; mkdir -p /tmp/sp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp/stats -o r -- python $R/profiles/experiments/small_launch_trace.py > /dev/null 2>&1
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/sp/pmc_$n -o r -- python $R/profiles/experiments/small_launch_trace.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = ["# rocprofv3 --kernel-trace --stats of profiles/experiments/small_launch_trace.py (200 single-capture calls, firmware shapes)", "kernel,calls,avg_ns,min_ns,max_ns"]
for f in glob.glob("/tmp/sp/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sr::" in r["Name"]:
            out.append(f"\"{r['Name']}\",{r['Calls']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']}")
out += ["", "# PMC passes (each its own run), mean per launch over the calls of the loop", "kernel,counter,mean_per_launch"]
for d in sorted(glob.glob("/tmp/sp/pmc_*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "sr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            out.append(f"\"{k}\",{c},{sum(v) / len(v):.1f}")
open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04_small_launch_rocprof.csv"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
