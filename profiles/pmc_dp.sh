#!/bin/bash
# usage: profiles/pmc_dp.sh <tag>   (on the GPU box, through gpurun)
# rocprofv3 passes for the opt-in full-DP scorer alone (bench.py --scorer dp, 65 536 x 100 pairs): kernel trace + stats,
# then one run per --pmc counter group; writes profiles/<tag>_dp_rocprof_summary.csv and profiles/pmc_valu_dp.json
# (issue slots per pair, with the sha of the kernel sources) and copies them to gpurun_out/keep_<tag>_dp/.
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$1_dp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $R/bench.py --scorer dp --steps 3 > $OUT/bench_under_rocprof.json 2>/dev/null
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o r -- python $R/bench.py --scorer dp --steps 1 > /dev/null 2>&1
done
cd $R
python - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
B, K = 65536, 100
out = [f"# rocprofv3 summary {tag} (full-DP scorer alone): python bench.py --scorer dp --steps 3 on 1x MI355X (B={B}, K={K}, T=256)",
       "kernel,calls,avg_ns,min_ns,max_ns,pct"]
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "sr::" in r["Name"]:
            out.append(f"\"{r['Name']}\",{r['Calls']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']},{r['Percentage']}")
vals = {}
out += ["", "# PMC passes (each its own run), last launch of the DP kernel", "kernel,counter,value"]
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_dtw_dp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = float(r["Counter_Value"])
    cyc = acc.get("GRBM_GUI_ACTIVE", 0) / 8.0
    for c, v in sorted(acc.items()):
        out.append(f"sr::k_dtw_dp_band,{c},{v:.0f}")
        vals[c] = (v, cyc)
out += ["", "# derived", "kernel,metric,value"]
if "SQ_ACTIVE_INST_VALU" in vals and "SQ_INSTS_VALU" in vals:
    sl, cyc = vals["SQ_ACTIVE_INST_VALU"]
    out.append(f"sr::k_dtw_dp_band,valu_issue_busy_fraction,{min(1.0, sl * 4.0 / (1024.0 * cyc)):.3f}")
    out.append(f"sr::k_dtw_dp_band,valu_slots_per_pair,{sl / (B * K):.1f}")
    out.append(f"sr::k_dtw_dp_band,valu_insts_per_pair,{vals['SQ_INSTS_VALU'][0] / (B * K):.1f}")
    if "SQ_WAVE_CYCLES" in vals:
        out.append(f"sr::k_dtw_dp_band,mean_waves_per_simd,{vals['SQ_WAVE_CYCLES'][0] * 4.0 / (1024.0 * vals['SQ_WAVE_CYCLES'][1]):.2f}")
    if "SQ_LDS_IDX_ACTIVE" in vals:
        out.append(f"sr::k_dtw_dp_band,lds_busy_fraction,{min(1.0, vals['SQ_LDS_IDX_ACTIVE'][0] / (256.0 * vals['SQ_LDS_IDX_ACTIVE'][1])):.3f}")
    json.dump({"source": f"profiles/{tag}_dp_rocprof_summary.csv", "B": B, "K": K, "kernel": "k_dtw_dp_band<8>",
               "k_dtw_dp_valu_slots_per_pair": sl / (B * K), "k_dtw_dp_valu_insts_per_pair": vals["SQ_INSTS_VALU"][0] / (B * K),
               "kernel_sources_sha": bench.kernel_sources_sha("pmc_valu_dp.json")}, open("profiles/pmc_valu_dp.json", "w"), indent=1)
open(f"profiles/{tag}_dp_rocprof_summary.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
mkdir -p $R/gpurun_out/keep_${TAG}_dp
cp profiles/${TAG}_dp_rocprof_summary.csv profiles/pmc_valu_dp.json $OUT/bench_under_rocprof.json $R/gpurun_out/keep_${TAG}_dp/ 2>/dev/null
rm -rf $OUT
