#!/usr/bin/env python
"""bench.py -- utterances/s of the isolated-word recognition hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one pass of the whole hot path (noise_atap -> VAD -> MFCC -> greedy DTW x K templates ->
argmin) over one batch of synthetic capture buffers that is already resident in HBM.
Workload at every N (weak scaling): BASELINE.json configs[2] per GPU -- 65 536 utterances of 256 frames
(25 360-sample 8 kHz capture buffers) x 100 templates, 12 MFCC coefficients; utterances are sharded
over ranks, templates replicated, and each step ends with one RCCL all-gather of the per-template score
matrix (N > 1 only; double-buffered so that it overlaps the next step's kernels).  Inside a step the engine cuts the
batch into chunks on three internal streams.  Prints ONE JSON line on rank 0.
`--batch 4096 --templates 10` is BASELINE configs[1] (a parity-test shape, not the headline metric).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stm32_speech_recognition_amd import Engine, synth  # noqa: E402
from stm32_speech_recognition_amd import dist_util as du  # noqa: E402
from stm32_speech_recognition_amd.engine import results_from_torch, vad_from_torch  # noqa: E402

T = 256            # frames per utterance (metric: "256-frame, 100 templates")
K = 100            # templates
N_WORDS = 20       # vocabulary the templates are spoken from (comm_num, Flash.H:17)
MAX_FRAMES = 320   # frame cap: templates run 192..320 frames
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_utt(S, T, C, K):
    """compulsory HBM traffic of the whole path per utterance (templates amortised to 0)."""
    return 2 * S + 2 * (2 * T * C) + 4 * K + 16


def mfcc_kernel_bytes_per_utt(T, C):
    """k_mfcc alone: speech span read once (+1 pre-emphasis halo sample), MFCC rows written once, 48 B record."""
    return 2 * (80 * (T - 1) + 160 + 1) + 2 * T * C + 48


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="utterances per GPU per step")
    ap.add_argument("--templates", type=int, default=None, help="templates (default: 100, the metric's config; 500 for ext)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="utterances timed on the host cores")
    ap.add_argument("--workload", choices=["ref", "ext"], default="ref",
                    help="ref = BASELINE configs[2] (the metric's config); ext = configs[4], the 16 kHz/512-pt/40-Mel x 500 "
                         "templates EXTENSION (no reference counterpart; not the headline metric)")
    args = ap.parse_args()
    global K, N_WORDS
    rate, eng_cfg = 1, {}
    if args.workload == "ext":
        rate, eng_cfg, K, N_WORDS = 2, dict(fs=16000, nfft=512, n_mel=40), 500, 100
    if args.templates:
        K = args.templates
        N_WORDS = min(N_WORDS, K)

    rank, local_rank, world = du.env_rank()
    # test hooks (tests/test_gpu_parity.py runs the N = 2 code path on a 1-GPU box): collective backend and a forced
    # device ordinal.  The driver never sets them: N > 1 means one rank per GPU over RCCL.
    backend = os.environ.get("SR_BENCH_BACKEND", "nccl")
    if "SR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SR_BENCH_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    dist = None
    if world > 1:
        import torch.distributed as dist
        du.init_process_group(backend, local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.batch
    S = synth.buf_len_for(T, rate)

    eng = Engine(max_frames=MAX_FRAMES, device=local_rank, **eng_cfg)

    # ---- templates: K synthetic words through the SAME front end (main.c:121-138 save_mdl) ----------
    bank = synth.word_bank(N_WORDS)
    rng = np.random.default_rng(2026)
    tfr = rng.integers(192, 321, K)
    tpcm = synth.make_utterances(np.arange(K) % N_WORDS, tfr, seed=77, bank=bank, S=synth.buf_len_for(320, rate), device=dev,
                                 rate=rate)
    tvad, tmf = eng.features_dev(tpcm)
    torch.cuda.synchronize()
    tv = vad_from_torch(tvad)
    assert (tv["status"] == 0).all() and np.array_equal(tv["frm_num"], tfr), "template front end did not yield the planned frame counts"
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    del tpcm, tvad, tmf

    # ---- this rank's shard of utterances, generated straight into HBM -------------------------------
    lo, hi = du.shard_bounds(world * B, world, rank)  # weak scaling: B utterances per rank
    words = torch.from_numpy(rng.integers(0, N_WORDS, world * B))[lo:hi]
    pcm = synth.make_utterances(words, [T] * B, seed=1000 + rank, bank=bank, S=S, device=dev, rate=rate)
    # N > 1: two output sets, so the all-gather of step i (RCCL stream) overlaps the kernels of step i+1
    outs = [eng.alloc_outputs(B, dev, mfcc=True, vad=True) for _ in range(2 if world > 1 else 1)]
    out = outs[0]
    xchg = du.ScoreExchange(world, [torch.empty(world * B, K, dtype=torch.int32, device=dev) for _ in outs]) if world > 1 else None
    n_step = [0]

    def step():
        j = n_step[0] % len(outs)
        n_step[0] += 1
        if xchg is not None:
            xchg.reserve(j)  # the gather that read outs[j]["scores"] two steps ago is ordered before the kernels
        eng.recognize_dev(pcm, outs[j])
        if xchg is not None:  # the path's one exchange step: all-gather of per-template scores over xGMI
            xchg.launch(j, outs[j]["scores"])

    def finish():
        if xchg is not None:
            xchg.drain()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish()
    eng.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()  # every step's kernels AND its all-gather are complete
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = du.max_over_ranks(dt, dev, world)
    stage = eng.stage_ms()  # hipEvent timings of the timed steps, each kernel on the stream it was launched on
    eng.set_profiling(False)
    # untimed extra: the same step as ONE chunk on one stream, for the per-kernel durations without overlap
    eng.set_pipeline(streams=1)
    eng.set_profiling(True)
    for _ in range(2):
        eng.recognize_dev(pcm, out)
    torch.cuda.synchronize()
    stage_iso = eng.stage_ms()
    eng.set_profiling(False)
    eng.set_pipeline()
    # sanity on real outputs (outside the timed region): every utterance must have exactly T frames
    res = results_from_torch(out["results"])
    assert (res["status"] == 0).all() and (res["frm_num"] == T).all(), "workload is not 256-frame utterances"
    acc = float((res["best_tpl"] % N_WORDS == words.numpy()).mean())
    if xchg is not None:  # the gathered matrix holds every rank's scores in global utterance order
        g = xchg.gathered[0][rank * B:(rank + 1) * B]
        assert torch.equal(g, out["scores"]), "all-gather did not return this rank's shard in place"

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        C = 12
        by_path = algorithmic_bytes_per_utt(S, T, C, K)
        by_mfcc = mfcc_kernel_bytes_per_utt(T, C) if rate == 1 else 2 * (160 * (T - 1) + 320 + 1) + 2 * T * C + 48
        # the engine cuts a step into chunks on overlapping streams: each kernel is launched `launches` times per step,
        # one launch covers B / launches utterances; stage[...] are per-launch durations (hipEvents on its stream)
        launches = stage["launches_per_call"]
        ach = by_mfcc * (B / launches) / (stage["mfcc"] * 1e-3) / 1e9
        # the same kernel alone on the chip: ONE launch over all B utterances (the untimed extra pass)
        ach_iso = by_mfcc * B / (stage_iso["mfcc"] * 1e-3) / 1e9
        traffic = traffic_iso = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.workload == "ref" and os.path.exists(tpath):
            try:  # the PMC passes measured one whole-batch launch over tj["B"] utterances; traffic scales with utterances
                tj = json.load(open(tpath))
                traffic = tj["k_mfcc_hbm_bytes_per_launch"] * (B / launches) / tj["B"]
                traffic_iso = tj["k_mfcc_hbm_bytes_per_launch"] * B / tj["B"]
                traffic_src = tj.get("source")
            except Exception:
                traffic = traffic_iso = None
        # What actually bounds the path: VALU issue.  Measured directly on this chip (profiles/r02/VALU_ISSUE.md,
        # profiles/valu_issue_ubench.hip): in a mixed instruction stream every wave64 VALU instruction holds its SIMD's
        # issue port for 4 cycles (a transcendental for 8); the 2-cycle rate of simple ops needs a pure run of them, which
        # these kernels never have.  Ceiling = 1024 SIMDs x 2.4 GHz / 4 issue slots per second; the slots a step needs come
        # from the committed PMC pass (SQ_ACTIVE_INST_VALU = 4-cycle issue slots, profiles/pmc_valu.json).
        roofline_valu = None
        vpath = os.path.join(ROOT, "profiles", "pmc_valu.json")
        if args.workload == "ref" and os.path.exists(vpath):
            try:
                vj = json.load(open(vpath))
                insts = sum(v for k, v in vj.items() if k.endswith("_valu_insts_per_utt"))
                slots = sum(v for k, v in vj.items() if k.endswith("_valu_slots_per_utt")) or insts
                peak = 1024 * 2.4e9 / 4.0
                achv = slots * B / (stage["total"] * 1e-3)
                roofline_valu = {"bound": "valu-issue", "achieved": achv, "peak": peak, "unit": "4-cycle issue slots/s",
                                 "frac": achv / peak, "valu_slots_per_utt": slots, "valu_insts_per_utt": insts,
                                 "cycles_per_slot": 4.0, "clock_hz_assumed": 2.4e9, "source": vj.get("source"),
                                 "rates_source": "profiles/r02/VALU_ISSUE.md (per-opcode s_memtime micro-benchmark)",
                                 "note": "derived: slot counts from the committed PMC pass x this run's step time; the chip "
                                         "clocks 2.3-2.4 GHz under this load, the ceiling assumes the nominal 2.4"}
            except Exception:
                roofline_valu = None
        line = {
            "metric": f"utterances/sec (256-frame, {K} templates)" if args.workload == "ref"
            else "utterances/sec (EXTENSION: 16 kHz/512-pt/40 Mel, 256-frame, 500 templates; no reference counterpart)",
            "value": value,
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{2 if (B, K) == (65536, 100) else 1 if (B, K) == (4096, 10) else '-'}]: "
                                    f"batch={B} utterances x {K} templates per GPU, 256 frames, "
                                    "12-coef MFCC, 8 kHz 25360-sample capture buffers") if args.workload == "ref" else
                                   "BASELINE configs[4] EXTENSION: 16 kHz / 512-pt / 40 Mel, 256 frames x 500 templates",
                       "batch_per_gpu": B, "templates": K, "frames": T, "buf_len": S,
                       "parallelism": f"utterance-sharded x{world}" + (", RCCL all-gather of scores" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "k_mfcc", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": by_mfcc * B / launches, "kernel_ms": stage["mfcc"],
                         "launches_per_step": launches, "utterances_per_launch": B / launches,
                         "isolated": {"achieved": ach_iso, "frac": ach_iso / HBM_PEAK_GBS, "kernel_ms": stage_iso["mfcc"],
                                      "utterances_per_launch": B, "algorithmic_bytes_per_launch": by_mfcc * B,
                                      "traffic": traffic_iso,
                                      "note": "the same kernel alone on the chip: one launch over the whole batch (untimed "
                                              "extra pass); this is the figure the rocprof 'whole batch' rows agree with"},
                         "note": "achieved/frac: launches of the TIMED steps (B/launches utterances each, overlapping two other "
                                 "chunks' kernels on other streams, so each launch owns only part of the chip); the path is "
                                 "integer-VALU-bound, not HBM-bound (DESIGN.md 3.2): see roofline_valu"},
            "roofline_path": {"bytes_per_utt": by_path, "achieved": by_path * B / (stage["total"] * 1e-3) / 1e9,
                              "unit": "GB/s", "frac": by_path * B / (stage["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "whole step (all chunks, fork -> join on the launch stream)"},
            "roofline_valu": roofline_valu,
            "kernel_ms": stage,
            "kernel_ms_isolated": {**stage_iso, "note": "untimed extra pass: whole batch as one chunk on one stream (no overlap)"},
            "top1_word_accuracy": acc,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pcm, eng, out, tm, tfr, args.cpu_sample, eng_cfg)
            if args.workload == "ref":
                line["cpu_reference_objects"] = cpu_reference_objects(local_rank)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def usable_cores():
    """host threads this process may really keep busy: min(affinity mask, cgroup CPU quota).  The GPU boxes expose 256
    hardware threads but run the job under a cgroup quota (cpu.max), and oversubscribing a quota only adds throttling."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        quota = None if q == "max" else int(q) / int(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())      # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except Exception:
            quota = None
    if quota:
        n = max(1, min(n, int(quota)))
    return n, (os.cpu_count() or n), quota


def cpu_baseline(pcm, eng, out, tm, tfr, n, eng_cfg):
    """The reference C path on the host cores, on the first n utterances of the timed batch.

    kind "reference" (reference workload): the reference's OWN VAD.C / MFCC.C / DTW.C objects (oracle/_ref/libsr_ref320.so:
    compiled from /root/reference where the sources lie, with the one compile-time constant that caps a record at 119
    frames raised to 320, oracle/Makefile) + the C transcription of the assembly FFT + the restated spch_recg scan.  The
    objects keep file-scope statics (MFCC.C:14-15, DTW.C:65-68), so each host thread dlopens its own private copy of the
    .so; ctypes drops the GIL during the calls.  kind "port": the parametrised restatement (oracle tier ii), used for the
    extension workload (no reference counterpart) or when the reference objects are absent; it is also timed as a side
    figure.  The GPU scores / argmin of the sampled utterances are cross-checked against the CPU results."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cores, hw_threads, quota = usable_cores()
    host = synth.as_u16_numpy(pcm[:n])
    gsc = out["scores"][:n].cpu().numpy().view(np.uint32)
    gres = results_from_torch(out["results"][:n])
    where = (f"first {n} utterances of the timed batch, {cores} host threads (box: {hw_threads} hardware threads, "
             f"cgroup CPU quota {quota if quota else 'none'})")
    # ---- port (tier ii), multi-threaded inside the C library
    orc = ol.Oracle(max_frames=MAX_FRAMES, **eng_cfg)
    tpl = orc.make_templates(tm, tfr.astype(np.uint32))
    orc.recognize_batch(host[:cores * 2], tpl, n_threads=cores, want_mfcc=False, want_scores=False)  # warm-up
    t0 = time.perf_counter()
    ores, _, osc = orc.recognize_batch(host, tpl, n_threads=cores, want_mfcc=False, want_scores=True)
    dt_port = time.perf_counter() - t0
    match_port = bool(np.array_equal(gsc, osc) and np.array_equal(gres["best_tpl"], ores["best_tpl"]))
    port = {"value": n / dt_port, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": where + ", gcc -O2 oracle (tier ii)", "seconds": dt_port,
            "gpu_results_identical_on_sample": match_port}
    if eng_cfg or not ol.RefLib320.available():
        return port
    # ---- the reference's own objects, one private copy of the .so per thread
    import ctypes as C
    import shutil
    import tempfile
    import threading
    Kt = len(tfr)
    stride = 8192
    store = np.full(Kt * stride, 0xFF, dtype=np.uint8)
    for k in range(Kt):                                     # v_ftr_tag images: save_sign | frm_num | mfcc_dat (MFCC.H:18-25)
        rec = store[k * stride:(k + 1) * stride]
        rec[:4].view(np.uint16)[:] = (12345, tfr[k])
        rec[4:4 + int(tfr[k]) * 24] = np.ascontiguousarray(tm[k, :tfr[k]]).view(np.uint8).reshape(-1)
    tmp = tempfile.mkdtemp(prefix="sr_ref_")
    libs = []
    for i in range(cores):
        pth = os.path.join(tmp, f"libsr_ref320_{i}.so")
        shutil.copyfile(ol.REF320_PATH, pth)
        libs.append(C.CDLL(pth))
    S = host.shape[1]
    r_sc = np.zeros((n, Kt), dtype=np.uint32)
    r_best = np.zeros(n, dtype=np.uint32)
    r_dis = np.zeros(n, dtype=np.uint32)
    r_st = np.zeros(n, dtype=np.int32)

    def work(i, lo, hi):
        L = libs[i]
        ftr = np.zeros(ol.RefLib320.FTR_BYTES, dtype=np.uint8)
        best, dis = C.c_uint32(0), C.c_uint32(0)
        for b in range(lo, hi):
            r_st[b] = L.sr_ref_spch_recg_seg(host[b].ctypes.data_as(C.c_void_p), C.c_uint16(S), C.c_uint16(2400),
                                             store.ctypes.data_as(C.c_void_p), C.c_uint32(Kt), C.c_uint32(stride),
                                             C.c_uint32(0), ftr.ctypes.data_as(C.c_void_p), C.byref(best), C.byref(dis),
                                             r_sc[b].ctypes.data_as(C.c_void_p))
            r_best[b], r_dis[b] = best.value, dis.value

    def run(n_run):
        per = (n_run + cores - 1) // cores
        th = [threading.Thread(target=work, args=(i, min(i * per, n_run), min((i + 1) * per, n_run))) for i in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    run(min(n, 2 * cores))                                  # warm-up
    dt = run(n)
    shutil.rmtree(tmp, ignore_errors=True)
    match = bool((r_st == 0).all() and np.array_equal(gsc, r_sc) and np.array_equal(gres["best_tpl"], r_best)
                 and np.array_equal(gres["min_dis"], r_dis))
    return {"value": n / dt, "unit": "utterances/s", "cores": cores, "kind": "reference",
            "sample": where + ", the reference's own VAD.C/MFCC.C/DTW.C objects (gcc -O2, vv_tim_max raised to 320 frames) "
                              "+ C transcription of the asm FFT, one private .so copy per thread",
            "seconds": dt, "gpu_results_identical_on_sample": match, "port": port}


def cpu_reference_objects(device, n=32, Kr=100):
    """Side figure: the reference's OWN objects (oracle/_ref: VAD.C / MFCC.C / DTW.C compiled verbatim + the C
    transcription of the assembly FFT) timed next to the port on the largest shape they can run -- their constants are
    compile-time, so 119 frames in a 16 000-sample capture (ADC.H:8, MFCC.H:15-16) -- single thread (file-scope
    statics make them non-reentrant).  Shows that the port used for `cpu_baseline` is not slower than the reference's
    code, and cross-checks the engine against the reference objects on these captures."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    if not ol.RefLib.available():
        return None
    T, S = 119, 16000
    ref, orc = ol.RefLib(), ol.Oracle(max_frames=T)
    eng = Engine(max_frames=T, device=device)
    bank = synth.word_bank(N_WORDS)
    rng = np.random.default_rng(7)
    tfr = rng.integers(60, T + 1, Kr)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(Kr) % N_WORDS, tfr, seed=5, bank=bank, S=S))
    store, st = eng.train_store(tp, np.arange(Kr), n_slots=Kr)       # save_mdl flash image (main.c:121-138)
    assert (st == 0).all()
    eng.set_templates_store(store)
    words = rng.integers(0, N_WORDS, n)
    pcm = synth.as_u16_numpy(synth.make_utterances(words, [T] * n, seed=6, bank=bank, S=S))
    tm = np.zeros((Kr, T + 1, 12), np.int16)
    for k in range(Kr):
        tm[k, :T] = store[k * 4096 + 4:k * 4096 + 4 + T * 24].view(np.int16).reshape(T, 12)
    tpl = orc.make_templates(tm, tfr.astype(np.uint32))
    # alternate the two single-thread runs and keep the best of three each: the cgroup CPU quota throttles whichever
    # leg happens to run right after the 16-thread cpu_baseline burst
    t_ref = t_port = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        r = [ref.spch_recg(pcm[b], store) for b in range(n)]               # (status, best, dis, scores, mfcc, n)
        t_ref = min(t_ref, time.perf_counter() - t0)
        t0 = time.perf_counter()
        ores, _, osc = orc.recognize_batch(pcm, tpl, n_threads=1, want_mfcc=False, want_scores=True)
        t_port = min(t_port, time.perf_counter() - t0)
    g = eng.recognize(pcm, want_mfcc=False, want_vad=False)
    same = all(r[b][0] == 0 and r[b][1] == g["results"]["best_tpl"][b] == ores["best_tpl"][b]
               and r[b][2] == g["results"]["min_dis"][b] and np.array_equal(r[b][3], g["scores"][b])
               and np.array_equal(r[b][3], osc[b]) for b in range(n))
    eng.close()
    return {"shape": f"{n} captures x {Kr} templates, {T} frames, 16000-sample buffers (the reference's compile-time limits)",
            "reference_objects_utt_per_s_1_thread": n / t_ref, "port_utt_per_s_1_thread": n / t_port,
            "engine_reference_port_identical": bool(same)}


if __name__ == "__main__":
    main()
