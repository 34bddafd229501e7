"""Turn rocprofv3 CSV output (kernel stats + separate --pmc passes) into the committed summaries.

  python profiles/summarize.py <gpurun_out/profdir> <tag> [batch] [description]
        ->  profiles/<tag>_rocprof_summary.csv, and for the reference workload (no description given)
            profiles/pmc_traffic.json + profiles/pmc_valu.json  (read by bench.py)
profdir layout: stats/*kernel_stats.csv, pmc_<COUNTER>/*counter_collection.csv (one dir per --pmc pass, every SQ pass
also carrying GRBM_GUI_ACTIVE) and traffic_<batch>_<COUNTER>/ (FETCH_SIZE / WRITE_SIZE at a smaller batch).
Only the engine's own kernels (sr::*) are kept; torch's data-generation kernels are dropped.
HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE
reports half of a coalesced stream, hence hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys


def sources_sha(which):
    """bench.py's kernel_sources_sha: the committed PMC figures carry the sha of the kernel sources they were taken with"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_sources_sha(which)


def main():
    root, tag = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    desc = sys.argv[4] if len(sys.argv) > 4 else None
    here = os.path.dirname(os.path.abspath(__file__))
    out = [f"# rocprofv3 summary {tag}: python bench.py --steps 5 --warmup 1 --no-cpu-baseline on 1x MI355X "
           + (desc if desc else f"(B={batch}, K=100, T=256)"),
           "# rocprofv3 --kernel-trace --stats --output-format csv   (all launches mixed: template pass, chunks, whole-batch pass)",
           "kernel,calls,avg_ns,min_ns,max_ns,pct"]
    for f in glob.glob(os.path.join(root, "stats", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "sr::" in r["Name"]:
                out.append(f"\"{r['Name']}\",{r['Calls']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']},{r['Percentage']}")
    # the same trace without the 100-utterance template pass (launches shorter than half the longest one are dropped):
    # this is the per-launch duration bench.py's roofline.kernel_ms has to agree with
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "stats", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "sr::" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((float(r["Start_Timestamp"]), float(r["End_Timestamp"])))
    out += ["", "# launches by class, from the kernel trace of the same run in dispatch order: the template pass (first",
            "# k_mfcc launch, 100 utterances; its VAD is a k_vad_wide launch, not listed) is dropped; 'chunk' = warm-up + timed steps (B/12 utterances per launch on three",
            "# streams, overlapping other chunks' kernels) -- the per-launch duration bench.py's kernel_ms has to agree with;",
            "# 'whole batch' = the last 2 launches = bench.py's untimed extra pass (one chunk, one stream)",
            "kernel,class,launches,avg_ns"]
    wide_vad = any("k_vad_wide" in k for k in per)  # the 100-utterance template pass takes the small-launch VAD form
    for k, v in per.items():
        d = [e - b for b, e in sorted(v)]
        if "k_vad_wide" in k:
            continue
        if ("k_vad" in k and not wide_vad) or "k_mfcc" in k:
            d = d[1:]
        if not d:
            continue
        chunk, whole = d[:-2], d[-2:]
        if chunk:
            out.append(f"\"{k}\",chunk,{len(chunk)},{sum(chunk) / len(chunk):.0f}")
        out.append(f"\"{k}\",whole batch,{len(whole)},{sum(whole) / len(whole):.0f}")
    out += ["", "# PMC passes (each its own run: rocprofv3 --kernel-trace --pmc <counters>), LAST launch = the whole-batch pass",
            "kernel,counter,value"]
    vals, cyc = {}, {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        fs = glob.glob(os.path.join(d, "*counter_collection.csv"))
        if not fs:
            continue
        acc = collections.defaultdict(dict)
        for r in csv.DictReader(open(fs[0])):
            if "sr::" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                acc[name][r["Counter_Name"]] = float(r["Counter_Value"])     # last launch wins = the whole-batch pass
        for k in acc:
            for c, v in sorted(acc[k].items()):
                if c != "GRBM_GUI_ACTIVE" or (k, c) not in vals:
                    out.append(f"{k},{c},{v:.0f}")
                vals[(k, c)] = v
                if "GRBM_GUI_ACTIVE" in acc[k]:
                    cyc[(k, c)] = acc[k]["GRBM_GUI_ACTIVE"] / 8.0      # shader cycles of THIS pass (counter sums the 8 XCDs)
    # derived.  SQ_ACTIVE_INST_VALU counts issue slots of 4 cycles ("quad-cycles"; a transcendental takes two): it equals
    # SQ_INSTS_VALU + the transcendentals, which is what profiles/r02/VALU_ISSUE.md measured directly.  Busy fraction =
    # slots * 4 / (1024 SIMDs * shader cycles of the same pass).
    out += ["", "# derived (1024 SIMDs, 256 CUs; each fraction uses the GRBM_GUI_ACTIVE / 8 shader cycles of the pass its counter came from)",
            "kernel,metric,value"]
    for k in sorted(set(kk for kk, _ in vals)):
        if k == "sr::k_argmin":
            continue
        try:
            c_v = cyc[(k, "SQ_ACTIVE_INST_VALU")]
            out.append(f"{k},shader_cycles,{c_v:.0f}")
            out.append(f"{k},valu_issue_busy_fraction,{min(1.0, vals[(k, 'SQ_ACTIVE_INST_VALU')] * 4.0 / (1024.0 * c_v)):.3f}")
            out.append(f"{k},issue_slots_per_valu_inst,{vals[(k, 'SQ_ACTIVE_INST_VALU')] / vals[(k, 'SQ_INSTS_VALU')]:.3f}")
            if (k, "SQ_LDS_IDX_ACTIVE") in vals:
                c_l = cyc[(k, "SQ_LDS_IDX_ACTIVE")]
                out.append(f"{k},lds_busy_fraction,{min(1.0, vals[(k, 'SQ_LDS_IDX_ACTIVE')] / (256.0 * c_l)):.3f}")
                out.append(f"{k},lds_conflict_fraction,{vals[(k, 'SQ_LDS_BANK_CONFLICT')] / max(1.0, vals[(k, 'SQ_LDS_IDX_ACTIVE')]):.3f}")
            if (k, "SQ_WAVE_CYCLES") in vals:
                out.append(f"{k},mean_waves_per_simd,{vals[(k, 'SQ_WAVE_CYCLES')] * 4.0 / (1024.0 * cyc[(k, 'SQ_WAVE_CYCLES')]):.2f}")
        except KeyError:
            pass
    for km in ("sr::k_mfcc", "sr::k_mfcc_ext"):
        if (km, "SQ_INSTS_VALU") in vals:
            out.append(f"{km},valu_insts_per_frame,{vals[(km, 'SQ_INSTS_VALU')] / (batch * 256.0):.1f}")
    if ("sr::k_dtw_lds", "SQ_INSTS_VALU") in vals:
        out.append(f"sr::k_dtw_lds,valu_insts_per_utt,{vals[('sr::k_dtw_lds', 'SQ_INSTS_VALU')] / float(batch):.0f}")
    # HBM traffic of k_mfcc: FETCH_SIZE / WRITE_SIZE passes at the smaller batch encoded in the directory name; the
    # last k_mfcc launch of the run is bench.py's whole-batch pass (all tb utterances in one launch)
    tr = {}
    for d in sorted(glob.glob(os.path.join(root, "traffic_*"))):
        tb = int(os.path.basename(d).split("_")[1])
        fs = glob.glob(os.path.join(d, "*counter_collection.csv"))
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if "sr::k_mfcc" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and not desc:
                tr[r["Counter_Name"]] = (float(r["Counter_Value"]), tb)
    if len(tr) == 2:
        (fs_, tb), (ws, _) = tr["FETCH_SIZE"], tr["WRITE_SIZE"]
        out += ["", f"# HBM traffic of the whole-batch k_mfcc launch at B = {tb} (separate --pmc passes; KB)",
                "kernel,counter,value", f"sr::k_mfcc,FETCH_SIZE,{fs_:.1f}", f"sr::k_mfcc,WRITE_SIZE,{ws:.1f}"]
        j = {"source": f"profiles/{tag}_rocprof_summary.csv", "kernel": "sr::k_mfcc", "B": tb,
             "FETCH_SIZE_KB": fs_, "WRITE_SIZE_KB": ws,
             "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM section; the factor 2 was "
                           "re-calibrated for this kernel's access pattern, profiles/r02/VALU_ISSUE.md)",
             "k_mfcc_hbm_bytes_per_launch": int((2 * fs_ + ws) * 1024), "kernel_sources_sha": sources_sha("pmc_traffic.json")}
        json.dump(j, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
    # VALU wave-instructions per utterance of the big kernels (whole-batch launch): bench.py prices a step against the VALU
    # issue ceiling with these.  One file per workload: pmc_valu.json (reference workload, B = 65536, K = 100),
    # pmc_valu_ext.json (description holds "--workload ext": configs[4]), pmc_valu_k10.json ("--templates 10": configs[1]),
    # pmc_valu_loud.json ("--gain 2.4": configs[2] at SURVEY.md 8(d)'s speech amplitudes)
    which = "pmc_valu.json" if not desc else "pmc_valu_ext.json" if "--workload ext" in desc else \
        "pmc_valu_k10.json" if "--templates 10" in desc else "pmc_valu_loud.json" if "--gain 2.4" in desc else None
    vj = {"source": f"profiles/{tag}_rocprof_summary.csv", "B": batch,
          "counter": "SQ_INSTS_VALU (wave-level instructions) and SQ_ACTIVE_INST_VALU (4-cycle issue slots: transcendentals count twice)"}
    for k in ("sr::k_vad", "sr::k_mfcc", "sr::k_mfcc_ext", "sr::k_dtw_lds", "sr::k_argmin"):
        if (k, "SQ_INSTS_VALU") in vals:
            vj[k.split("::")[1] + "_valu_insts_per_utt"] = vals[(k, "SQ_INSTS_VALU")] / float(batch)
        if (k, "SQ_ACTIVE_INST_VALU") in vals:
            vj[k.split("::")[1] + "_valu_slots_per_utt"] = vals[(k, "SQ_ACTIVE_INST_VALU")] / float(batch)
    if "sr::k_mfcc" in set(kk for kk, _ in vals) and ("sr::k_mfcc", "SQ_INSTS_VALU") in vals:
        vj["k_mfcc_valu_insts_per_frame"] = vals[("sr::k_mfcc", "SQ_INSTS_VALU")] / (batch * 256.0)
    if len(vj) > 3 and which and (desc or batch == 65536):
        vj["kernel_sources_sha"] = sources_sha(which)
        json.dump(vj, open(os.path.join(here, which), "w"), indent=1)
    open(os.path.join(here, f"{tag}_rocprof_summary.csv"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[-24:]))


if __name__ == "__main__":
    main()
