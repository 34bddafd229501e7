#!/bin/bash
# Counters of the two big kernels for each of several builds, on ONE box (run through gpurun):
#   profiles/experiments/pmc_ab.sh OUTNAME "lib1.so lib2.so ..." [batch] [extra ab.py flags, e.g. "--gain 2.4"]
# One rocprofv3 --pmc pass per counter group and library (never combined with a trace domain other than --kernel-trace);
# the LAST launch of each kernel in a pass is ab.py's whole-batch, one-stream launch.  Prints and stores
# gpurun_out/OUTNAME.txt: per library and kernel the counters, VALU instructions per frame / per pair, LDS conflict fraction
# (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) and issue-busy (SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES-normalised cycles).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; NAME=$1; LIBS=$2; BATCH=${3:-16384}; XF=${4:-}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/$NAME.txt
for lib in $LIBS; do
  for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d" " -f1); d=/tmp/pmc_ab_$$/$(basename $lib)_$n
    SR_ENGINE_LIB=$R/$lib timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- \
        python $R/profiles/experiments/ab.py x --child --batch $BATCH --steps 1 $XF > $OUT/$NAME.last.log 2>&1
    python - $d $(basename $lib) $BATCH >> $OUT/$NAME.txt <<'PY'
import csv, glob, sys, collections
d, lib, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_mfcc" in k or "k_dtw_lds" in k:
            acc["k_mfcc" if "k_mfcc" in k else "k_dtw_lds"][r["Counter_Name"]] = float(r["Counter_Value"])  # last launch wins
for k, v in acc.items():
    extra = ""
    if "SQ_INSTS_VALU" in v:
        extra = f" valu_per_frame={v['SQ_INSTS_VALU'] / (B * 256):.1f}" if k == "k_mfcc" else f" valu_per_utt={v['SQ_INSTS_VALU'] / B:.0f}"
    if "SQ_LDS_IDX_ACTIVE" in v and v["SQ_LDS_IDX_ACTIVE"]:
        extra += f" lds_conflict_fraction={v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}"
    if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_BUSY_CYCLES"):
        extra += f" lds_idx_active_per_busy={v['SQ_LDS_IDX_ACTIVE'] / v['SQ_BUSY_CYCLES']:.3f}"
    print(lib, k, " ".join(f"{c}={x:.0f}" for c, x in sorted(v.items())), extra)
PY
  done
done
rm -rf /tmp/pmc_ab_$$
cat $OUT/$NAME.txt
