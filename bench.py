#!/usr/bin/env python
"""bench.py -- utterances/s of the isolated-word recognition hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (plain command at every N: for N > 1 it launches its own ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

N > 1 without a launcher (WORLD_SIZE unset): `--launcher ranks` (default) starts N copies of this script, one rank per
GPU over RCCL (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set here, free port on 127.0.0.1), forwards rank 0's JSON line
as the only stdout line and returns the first non-zero exit code; `--launcher single` runs ONE process that drives the
N GPUs through the C ABI's sr_multi_* surface (csrc/sr_multi.cpp: one engine per device, one grouped in-place
ncclAllGather per step).  Under torch.distributed.run (WORLD_SIZE set) the script is one rank, as before.

A "step" = one pass of the whole hot path (noise_atap -> VAD -> MFCC -> greedy DTW x K templates ->
argmin) over one batch of synthetic capture buffers that is already resident in HBM.
Workload at every N (weak scaling): BASELINE.json configs[2] per GPU -- 65 536 utterances of 256 frames
(25 360-sample 8 kHz capture buffers) x 100 templates, 12 MFCC coefficients; utterances are sharded
over ranks, templates replicated, and each step ends with one RCCL all-gather of the per-template score
matrix (N > 1 only; double-buffered so that it overlaps the next step's kernels).  Inside a step the engine cuts the
batch into chunks on three internal streams.  Prints ONE JSON line on rank 0.
`--batch 4096 --templates 10` is BASELINE configs[1] (a parity-test shape, not the headline metric).
At N = 1 the default run also times BASELINE configs[1] and configs[4] (the extension front end, no reference
counterpart) for a few steps each and reports them under `other_configs` -- never in `value`.
The line carries `roofline` (dominant kernel against HBM as mandated, plus the fractions of the TIMED steps:
`timed_step_frac`, `valu_frac`, `valu_frac_at_measured_clock`), `roofline_valu` (the binding ceiling, per workload shape from
the committed PMC files) and `cpu_baseline` (the reference's own objects on the host cores, `cores_1` = one thread) with a
parity flag; at N > 1 the baseline sample covers the first utterances of EVERY rank's shard, the GPU side read back from
the gathered score matrix (`cpu_baseline.per_rank_sample`).
"""
import argparse
import json
import os
import socket
import subprocess
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
K_DEFAULT = 100    # templates
N_WORDS_DEFAULT = 20  # vocabulary the templates are spoken from (comm_num, Flash.H:17)
MAX_FRAMES = 320   # frame cap: templates run 192..320 frames
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_utt(S, T, C, K):
    """compulsory HBM traffic of the whole path per utterance (templates amortised to 0)."""
    return 2 * S + 2 * (2 * T * C) + 4 * K + 16


def mfcc_kernel_bytes_per_utt(T, C):
    """k_mfcc alone: speech span read once (+1 pre-emphasis halo sample), MFCC rows written once, 48 B record."""
    return 2 * (80 * (T - 1) + 160 + 1) + 2 * T * C + 48


def parse_args(argv=None):
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
    ap.add_argument("--launcher", choices=["ranks", "single"], default="ranks",
                    help="N > 1 started as a plain command: ranks = one process per GPU over torch.distributed/RCCL "
                         "(started by this script); single = one process, sr_multi_* C ABI")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra N = 1 measurements of configs[1] and configs[4] (other_configs)")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps of each other_configs entry")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the contract's N = 1/2/4/8 runs): --batch utterances PER GPU; strong: --batch is the "
                         "GLOBAL batch, split evenly over the GPUs (e.g. --batch 524288 = BASELINE configs[3] at every N)")
    ap.add_argument("--scorer", choices=["greedy", "dp"], default="greedy",
                    help="greedy = the reference's dtw() (the metric); dp = time ONLY the opt-in NON-REFERENCE full-DP scorer "
                         "(sr_dtw_dp_batch_dev) on the same features -- a side measurement, never the headline metric")
    ap.add_argument("--dp-lanes", type=int, default=0, help="--scorer dp: lanes per pair (0 default = 8; 4, 8, 16; 1 = first version)")
    ap.add_argument("--gain", type=float, default=1.0,
                    help="speech amplitude factor of the synthetic captures (synth.make_utterances): 1.0 = sinusoids of amplitude 64-300 "
                         "ADC codes (the headline since round 1), 2.4 = SURVEY.md 8(d)'s 200-600 (154-720).  The frame kernel's cost "
                         "depends on it (three magnitude / filterbank tiers, DESIGN.md 3.2), so every line carries `workload_stats`")
    ap.add_argument("--exchange", choices=["scores", "results"], default="scores",
                    help="N > 1: what the one collective of a step gathers -- the u32 score matrix [B, K] (north_star; 26 MB per rank "
                         "at K = 100) or the 16-byte result records [B] (1 MB per rank; SURVEY.md 8(e) names both)")
    ap.add_argument("--other-scale", type=int, default=1,
                    help="tests: divide the other_configs batches by this and run them whatever the headline shape is")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` as a plain command: start the N ranks ourselves (what torch.distributed.run would do:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, one process per GPU) and forward rank
    0's JSON line as the only line on stdout.  Everything else the ranks print goes to stderr.  The first rank that fails
    ends the job: the others are terminated (by PID) and its exit code is returned."""
    n = args.gpus
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SR_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    import signal
    import threading

    def stop_ranks(signum, frame):  # the job is being stopped from outside: take the ranks down with it (by PID)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.exit(128 + signum)

    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, stop_ranks)
    lines = []

    def pump():
        for ln in procs[0].stdout:
            if ln.startswith("{"):
                lines.append(ln)
            else:
                sys.stderr.write(ln)

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = 0
    alive = set(range(n))
    while alive and rc == 0:
        for r in list(alive):
            c = procs[r].poll()
            if c is not None:
                alive.discard(r)
                if c != 0:
                    rc = c
                    sys.stderr.write(f"bench.py: rank {r} exited with code {c}; stopping the other ranks\n")
        time.sleep(0.05)
    for r in alive:  # only after a failure
        procs[r].terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    th.join(timeout=10)
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n")
        rc = 1
    if rc == 0:
        j = json.loads(lines[0])
        j["launcher"] = "ranks (started by bench.py itself: one process per GPU over torch.distributed)"
        print(json.dumps(j), flush=True)
    return rc


def workload_setup(workload, Kt):
    """front-end configuration of a workload: (rate, engine config keywords, templates, vocabulary size)"""
    if workload == "ext":
        return 2, dict(fs=16000, nfft=512, n_mel=40), Kt or 500, min(100, Kt or 500)
    return 1, {}, Kt or K_DEFAULT, min(N_WORDS_DEFAULT, Kt or K_DEFAULT)


def make_templates(eng, bank, Kt, n_words, rate, dev, gain=1.0):
    """Kt synthetic words through the SAME front end (main.c:121-138 save_mdl) -> dense template set on the host"""
    rng = np.random.default_rng(2026)
    tfr = rng.integers(192, 321, Kt)
    tpcm = synth.make_utterances(np.arange(Kt) % n_words, tfr, seed=77, bank=bank, S=synth.buf_len_for(320, rate), device=dev,
                                 rate=rate, gain=gain)
    tvad, tmf = eng.features_dev(tpcm)
    torch.cuda.synchronize(dev)
    tv = vad_from_torch(tvad)
    assert (tv["status"] == 0).all() and np.array_equal(tv["frm_num"], tfr), "template front end did not yield the planned frame counts"
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((Kt, 1, 12), np.int16)], 1)
    return tm, tfr, rng


class ClockSampler:
    """shader clock of THIS process's GPU during the timed region, from sysfs (/sys/class/drm/cardN/device/pp_dpm_sclk: the
    line marked `*` is the current level; on MI300-class parts level 1 carries the live frequency).  The card is found by
    the PCI address HIP reports for the device (sysfs lists every GPU of the node, also the ones other tenants are using).
    Sampled every 20 ms on a host thread; summary() is None when the card or the file cannot be found."""

    def __init__(self, period=0.02, device=None):
        import glob
        import threading
        self.files = []
        try:
            dev = torch.cuda.current_device() if device is None else device
            pr = torch.cuda.get_device_properties(dev)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
                if os.path.basename(os.path.realpath(os.path.dirname(f))).lower().startswith(want):
                    self.files = [f]
        except Exception:
            self.files = []
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.files else None

    def _read(self):
        best = None
        for f in self.files:
            try:
                for ln in open(f).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        mhz = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                        best = mhz if best is None or mhz > best else best
            except Exception:
                pass
        return best

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v:
                self.samples.append(v)
            self._stop.wait(self.period)

    def __enter__(self):
        if self._th:
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._th:
            self._th.join(timeout=1)

    def summary(self):
        if not self.samples:
            return None
        a = np.array(self.samples)
        return {"mean_mhz": float(a.mean()), "min_mhz": float(a.min()), "max_mhz": float(a.max()), "samples": int(a.size),
                "source": f"sysfs {self.files[0]}, {int(self.period * 1e3)} ms period, during the timed region"}


FORCE_DIST = False  # test hook SR_BENCH_FORCE_DIST=1: initialise the process group and run the exchange even at N = 1


def measure(workload, B, Kt, steps, warmup, rank, world, local_rank, dist, gain=1.0, exchange="scores"):
    """K timed steps of one workload on this rank's GPU (barrier + synchronize on both sides, max over ranks), then an
    untimed isolated pass (whole batch as one chunk on one stream) for the per-kernel durations.  Returns a dict."""
    rate, eng_cfg, Kt, n_words = workload_setup(workload, Kt)
    dev = torch.device("cuda", local_rank)
    S = synth.buf_len_for(T, rate)
    eng = Engine(max_frames=MAX_FRAMES, device=local_rank, **eng_cfg)
    bank = synth.word_bank(n_words)
    tm, tfr, rng = make_templates(eng, bank, Kt, n_words, rate, dev, gain)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))

    # ---- this rank's shard of utterances, generated straight into HBM -------------------------------
    lo, hi = du.shard_bounds(world * B, world, rank)  # weak scaling: B utterances per rank
    words = torch.from_numpy(rng.integers(0, n_words, world * B))[lo:hi]
    pcm = synth.make_utterances(words, [T] * B, seed=1000 + rank, bank=bank, S=S, device=dev, rate=rate, gain=gain)
    # N > 1: two output sets, so the all-gather of step i (RCCL stream) overlaps the kernels of step i+1
    use_dist = dist is not None
    outs = [eng.alloc_outputs(B, dev, mfcc=True, vad=True) for _ in range(2 if use_dist else 1)]
    out = outs[0]
    # the path's one exchange step gathers the score matrix (north_star) or, --exchange results, the 16-byte result records
    xkey, xcols = ("scores", Kt) if exchange == "scores" else ("results", 4)
    xchg = du.ScoreExchange(world, [torch.empty(world * B, xcols, dtype=torch.int32, device=dev) for _ in outs],
                            force=FORCE_DIST) if use_dist else None
    n_step = [0]

    def step():
        j = n_step[0] % len(outs)
        n_step[0] += 1
        if xchg is not None:
            xchg.reserve(j)  # the gather that read outs[j]["scores"] two steps ago is ordered before the kernels
        eng.recognize_dev(pcm, outs[j])
        if xchg is not None:  # the path's one exchange step: all-gather of per-template scores (or result records) over xGMI
            xchg.launch(j, outs[j][xkey])

    def finish():
        if xchg is not None:
            xchg.drain()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    finish()
    eng.set_profiling(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    with ClockSampler() as clk:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        finish()  # every step's kernels AND its all-gather are complete
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
    dt_rank = dt
    dt = du.max_over_ranks(dt, dev, 2 if (use_dist and world == 1) else world)
    stage = eng.stage_ms()  # hipEvent timings of the timed steps, each kernel on the stream it was launched on
    eng.set_profiling(False)
    exchange_kind, exchange = exchange, None
    if use_dist:
        # diagnostics of the path's one exchange step (outside the timed region): every rank's own wall time of the timed
        # steps, the all-gather ALONE on an otherwise idle GPU (hipEvents on the current stream around a synchronous
        # call: the stream waits for the collective), and how much of it a step does not hide behind the next step's
        # kernels (wall time per step minus the fork -> join time of the engine's kernels; includes host launch gaps)
        per_rank = du.gather_floats(dt_rank / steps * 1e3, dev, world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ag = []
        for _ in range(3):
            dist.barrier()
            torch.cuda.synchronize()
            e0.record()
            dist.all_gather_into_tensor(xchg.gathered[0], outs[0][xkey].contiguous())
            e1.record()
            torch.cuda.synchronize()
            ag.append(e0.elapsed_time(e1))
        exchange = {"backend": dist.get_backend(), "ranks_in_communicator": dist.get_world_size(),
                    "gathers": "score matrix u32 [B, K]" if exchange_kind == "scores" else "result records, 16 bytes per utterance",
                    "step_ms_per_rank": per_rank, "allgather_ms": float(np.median(ag)),
                    "allgather_bytes_per_rank_out": int(world * B * xcols * 4),
                    "exposed_allgather_ms": max(0.0, dt / steps * 1e3 - stage["total"]),
                    "note": "allgather_ms: the collective alone (idle GPU, median of 3); exposed: wall time per step minus the "
                            "engine's fork->join kernel time of a step = what the double-buffered exchange does not hide "
                            "(upper bound: includes host launch gaps)"}
        try:
            exchange["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
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
    acc = float((res["best_tpl"] % n_words == words.numpy()).mean())
    if xchg is not None:  # the gathered matrix holds every rank's scores in global utterance order
        g = xchg.gathered[0][rank * B:(rank + 1) * B]
        assert torch.equal(g, out[xkey].view(B, xcols)), "all-gather did not return this rank's shard in place"
    sample_pcm = None
    if xchg is not None:  # first PER_RANK_PARITY_N capture buffers of every rank's shard, for rank 0's CPU parity check
        ns = min(PER_RANK_PARITY_N, B)
        sample_pcm = torch.empty(dist.get_world_size() * ns, S, dtype=pcm.dtype, device=dev)
        # as bytes: the 16-bit sample type is not a collective dtype of every backend (gloo refuses it)
        dist.all_gather_into_tensor(sample_pcm.view(torch.uint8), pcm[:ns].contiguous().view(torch.uint8))
    return dict(dt=dt, stage=stage, stage_iso=stage_iso, acc=acc, eng=eng, pcm=pcm, out=out, tm=tm, tfr=tfr, S=S, rate=rate,
                eng_cfg=eng_cfg, K=Kt, n_words=n_words, sclk=clk.summary(), exchange=exchange, gain=gain, exchange_kind=xkey,
                gathered=xchg.gathered[0] if xchg is not None else None, sample_pcm=sample_pcm)


SURVEY_GAIN = 2.4        # synth.make_utterances gain that gives SURVEY.md 8(d)'s sinusoid amplitudes (200-600: 64-300 x 2.4 = 154-720)
WORKLOAD_STATS_N = 32    # capture buffers whose frames are sorted into the frame kernel's tiers (host-side diagnostic)


def workload_stats(host, eng_cfg, gain):
    """What the timed workload looks like to the frame kernel (host-side diagnostic on the first captures of the batch; part of
    the CPU legs, absent with --no-cpu-baseline): the speech amplitude the generator was asked for and the share of frames in
    each of k_mfcc's magnitude / filterbank tiers (QUIET = every re^2 + im^2 of the frame <= 26 843, MID <=
    70 171, LOUD beyond), counted on the spectra of the CPU oracle (the product is not involved).  The extension front end has
    one code path, its fractions are reported for comparison only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    orc = ol.Oracle(max_frames=MAX_FRAMES, **eng_cfg)
    t = orc.frame_tiers(host)
    return {"gain": gain, "speech_amplitude": [round(64 * gain, 1), round(300 * gain, 1)],
            "speech_amplitude_note": "per-sinusoid amplitude range in ADC codes (three chirped sinusoids x envelope 0.55-1.0 + N(0,30)); "
                                     "SURVEY.md 8(d) specifies 200-600 = gain 2.4",
            "quiet_frame_fraction": t["quiet"], "mid_frame_fraction": t["mid"], "loud_frame_fraction": t["loud"],
            "frames_counted": t["frames"], "captures_counted": int(len(host)),
            "tiers_apply": not eng_cfg}


def workload_name(workload, B, Kt, gain=1.0):
    amp = "" if gain == 1.0 else (f", speech amplitude x{gain:g}" + (" = SURVEY.md 8(d)'s 200-600" if gain == SURVEY_GAIN else ""))
    if workload == "ext":
        return (f"BASELINE configs[4] EXTENSION (no reference counterpart): 16 kHz / 512-pt / 40 Mel, batch={B} utterances x "
                f"{Kt} templates per GPU, 256 frames" + amp)
    idx = 2 if (B, Kt) == (65536, 100) else 1 if (B, Kt) == (4096, 10) else "-"
    return (f"BASELINE configs[{idx}]: batch={B} utterances x {Kt} templates per GPU, 256 frames, 12-coef MFCC, 8 kHz "
            "25360-sample capture buffers" + amp)


def mfcc_bytes(rate):
    C = 12
    return mfcc_kernel_bytes_per_utt(T, C) if rate == 1 else 2 * (160 * (T - 1) + 320 + 1) + 2 * T * C + 48


def other_config(workload, B, Kt, steps, local_rank, cpu_n, gain=1.0):
    """one `other_configs` entry: a parity-test shape of BASELINE.json timed for a few steps under the same rules as the
    headline (inputs resident, barrier-free at N = 1, synchronize on both sides) + a CPU parity check on a sample"""
    m = measure(workload, B, Kt, steps, 1, 0, 1, local_rank, None, gain=gain)
    by = mfcc_bytes(m["rate"])
    iso = m["stage_iso"]
    e = {"workload": workload_name(workload, B, m["K"], gain), "value": B * steps / m["dt"], "unit": "utterances/s",
         "ms_per_step": m["dt"] / steps * 1e3, "steps": steps, "warmup": 1,
         "kernel_ms_isolated": {k: iso[k] for k in ("vad", "mfcc", "dtw", "argmin", "total")},
         "roofline_hbm_frac_dominant_kernel": by * B / (iso["mfcc"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
         "top1_word_accuracy": m["acc"]}
    rv, rv_stale = valu_roofline(valu_pmc_file(workload, B, m["K"], gain), B, m["stage"]["total"], m.get("sclk"))
    e["roofline_valu"], e["roofline_valu_stale"] = rv, rv_stale
    e["sclk"] = m.get("sclk")
    if cpu_n:
        e["workload_stats"] = workload_stats(synth.as_u16_numpy(m["pcm"][:WORKLOAD_STATS_N]), m["eng_cfg"], gain)
        cb = cpu_baseline(m["pcm"], m["eng"], m["out"], m["tm"], m["tfr"], min(cpu_n, B), m["eng_cfg"])
        e["parity_on_sample"] = {"identical": bool(cb["gpu_results_identical_on_sample"] and
                                                   cb.get("port", cb)["gpu_results_identical_on_sample"]),
                                 "utterances": min(cpu_n, B), "checker": cb["kind"] +
                                 (" (the reference's own objects) + port" if cb["kind"] == "reference" else
                                  " (the parametrised CPU restatement; this front end has no reference counterpart)"),
                                 "cpu_utt_per_s": cb["value"], "cpu_cores": cb["cores"]}
    if workload == "ext":
        e["note"] = "EXTENSION: no reference counterpart (cr4_fft_1024_stm32.s:214-215 converts 1024 points only); parity is against its own oracle"
    m["eng"].close()
    return e


def dp_cells_per_pair(in_n, tfr):
    """cells of the dtw_limit parallelogram (DTW.C:76-109) summed over the templates that pass the length gate, for
    utterances of in_n frames: the work of the full-DP scorer, counted with the same interval form the kernel uses"""
    tot, pairs = 0, 0
    for m in (int(v) for v in tfr):
        if in_n > 2 * m or 2 * in_n < m:
            continue
        X1, X2 = int((2 * m - in_n) / 3), int((4 * in_n - 2 * m) / 3)
        x = np.arange(1, in_n + 1)
        ub = np.where(x < X1, 2 * x + 1, ((x + 5 - in_n + 2 * m) >> 1) - 1)
        lb = np.where(x < X2, x >> 1, 2 * x + m - 2 * in_n - 3)
        tot += int(np.maximum(0, np.minimum(ub, m) - np.maximum(lb, 1) + 1).sum())
        pairs += 1
    return tot / max(pairs, 1), pairs


def measure_dp(B, Kt, steps, lanes, local_rank, parity_n=64):
    """the opt-in NON-REFERENCE full-DP scorer alone: features of B synthetic 256-frame utterances (computed once, resident
    in HBM) against Kt templates, `steps` timed launches of sr_dtw_dp_batch_dev; parity of a sample against its own oracle"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    dev = torch.device("cuda", local_rank)
    eng = Engine(max_frames=MAX_FRAMES, device=local_rank)
    bank = synth.word_bank(N_WORDS_DEFAULT)
    tm, tfr, rng = make_templates(eng, bank, Kt, N_WORDS_DEFAULT, 1, dev)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    eng.set_dp_lanes(lanes)
    vads, mfs = [], []
    for b0 in range(0, B, 16384):  # features chunk by chunk: the capture buffers are not needed afterwards
        n = min(16384, B - b0)
        pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, N_WORDS_DEFAULT, n)), [T] * n, seed=3000 + b0, bank=bank,
                                    S=synth.buf_len_for(T), device=dev)
        v, m = eng.features_dev(pcm)
        torch.cuda.synchronize()
        vads.append(v)
        mfs.append(m)
        del pcm
    vad, mfcc = torch.cat(vads), torch.cat(mfs)
    del vads, mfs
    sc = torch.empty(B, Kt, dtype=torch.int32, device=dev)
    eng.dtw_dp_dev(mfcc, sc, vad=vad)  # warm-up
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        eng.dtw_dp_dev(mfcc, sc, vad=vad)
        ev[i + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    cells, pairs_ok = dp_cells_per_pair(T, tfr)
    n = min(parity_n, B)
    parity = None
    if n:
        orc = ol.Oracle(max_frames=MAX_FRAMES)
        cores = usable_cores()[0]
        t1 = time.perf_counter()
        want = orc.dtw_dp_batch(mfcc[:n].cpu().numpy(), np.full(n, T, np.uint32), tm, tfr.astype(np.uint32), n_threads=cores)
        cpu_s = time.perf_counter() - t1
        same = bool(np.array_equal(sc[:n].cpu().numpy().view(np.uint32), want))
        parity = {"identical": same, "pairs": n * Kt, "checker": "sr_oracle_dtw_dp (own definition; no reference counterpart)",
                  "cpu_pairs_per_s": n * Kt / cpu_s, "cpu_cores": cores}
    name = {0: "k_dtw_dp_band<8>", 4: "k_dtw_dp_band<4>", 8: "k_dtw_dp_band<8>", 16: "k_dtw_dp_band<16>", 1: "k_dtw_dp_wave64"}[lanes]
    e = {"workload": f"OPT-IN NON-REFERENCE full-DP scorer alone: {B} utterances (256 frames) x {Kt} templates (192..320 frames), "
                     "features resident in HBM", "kernel": name, "lanes_per_pair": lanes or 8,
         "value": B * steps / dt, "unit": "utterances/s", "pairs_per_s": B * Kt * steps / dt, "ms_per_step": dt / steps * 1e3,
         "kernel_ms": float(np.mean(kms)), "steps": steps, "warmup": 1,
         "cells_per_pair": cells, "cells_per_s": B * pairs_ok * cells * steps / dt,
         "parity_on_sample": parity,
         "note": "the reference's dtw() is a greedy walk (DTW.C:150-188); this scorer never backs dtw() or the recognition path"}
    vpath = os.path.join(ROOT, "profiles", "pmc_valu_dp.json")
    if os.path.exists(vpath) and lanes in (0, 8):
        try:
            vj = json.load(open(vpath))
            if pmc_is_current(vj, "pmc_valu_dp.json"):
                slots = vj["k_dtw_dp_valu_slots_per_pair"]
                ach = slots * B * Kt / (float(np.mean(kms)) * 1e-3)
                e["roofline_valu"] = {"bound": "valu-issue", "achieved": ach, "peak": 1024 * 2.4e9 / 4.0, "frac": ach / (1024 * 2.4e9 / 4.0),
                                      "unit": "4-cycle issue slots/s", "valu_slots_per_pair": slots, "source": vj.get("source")}
        except Exception:
            pass
    eng.close()
    return e


def latency_block(local_rank, n_utt=64):
    """Side figure (never part of `value`): what ONE call of the drop-in symbols costs, next to the reference's own objects
    on one host core.  The firmware's shapes: 16 000-sample capture (ADC.H:8-9), 119-frame cap, an 80-slot store
    (comm_num * ftr_per_comm, Flash.H:15-17) trained through the same front end (main.c:121-138).
      spch_recg      one capture -> label + distance (main.c:249-296): upload, VAD, MFCC, DTW x 80, argmin, read-back
      get_mfcc       one segment -> v_ftr_tag (MFCC.C:86-191)
      dtw slot scan  main.c:279-291's loop: 80 dtw() calls for one input record (the symbol scores the record against every
                     cached model in ONE launch, the other 79 calls are look-ups)
      sr_recognize_batch_dev at B = 1, 16, 256 on device-resident captures, synchronised after every call
    Launches this small take the engine's small-launch forms (k_vad_wide, 4-frame tiles, k_dtw_cells; sr_set_small_launch);
    `*_batch_kernels_us` is the same call with them switched off."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from stm32_speech_recognition_amd import compat
    Tl, S, Kl = 110, 16000, 80
    dev = torch.device("cuda", local_rank)
    eng = Engine(max_frames=119, device=local_rank)
    bank = synth.word_bank(N_WORDS_DEFAULT)
    rng = np.random.default_rng(4)
    tfr = rng.integers(70, 120, Kl)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(Kl) % N_WORDS_DEFAULT, tfr, seed=8, bank=bank, S=S))
    store, st = eng.train_store(tp, np.arange(Kl), n_slots=Kl)
    assert (st == 0).all()
    eng.set_templates_store(store)
    compat.set_templates(store)
    words = rng.integers(0, N_WORDS_DEFAULT, n_utt)
    pcm = synth.as_u16_numpy(synth.make_utterances(words, [Tl] * n_utt, seed=9, bank=bank, S=S))
    ref = ol.RefLib() if ol.RefLib.available() else None

    p90 = {}

    def us(fn, n, key=None):
        """median microseconds per call over n calls (one stray call -- an allocation, a clock ramp -- must not set the
        figure); the 90th percentile is kept under `p90_us`"""
        fn(0)  # warm-up (first call creates the implicit engine / uploads the store)
        ts = []
        for i in range(n):
            t0 = time.perf_counter()
            fn(i)
            ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort()
        if key:
            p90[key] = ts[min(len(ts) - 1, int(0.9 * len(ts)))]
        return ts[len(ts) // 2]

    out = {"shape": f"{S}-sample captures, {Tl}-frame words, {Kl}-slot store, 119-frame cap (the firmware's constants)",
           "unit": "MEDIAN microseconds per call, host wall clock, one call in flight (python ctypes loop: ~2 us of interpreter "
                   "per call); 90th percentiles under p90_us"}
    # ---- spch_recg
    g = [None] * n_utt
    r = [None] * n_utt

    def f_spch(i):
        g[i] = compat.spch_recg(pcm[i])
    with ClockSampler(period=0.002) as clk:  # what the shader clock does under one-call-at-a-time load
        out["spch_recg_us"] = us(f_spch, n_utt, "spch_recg")
    out["sclk_during_spch_recg"] = clk.summary()
    # the same call with the small-launch forms switched off (sr_set_small_launch mode 1: one wave per capture in VAD, 64 frames
    # per frame-kernel workgroup, the batch DTW kernel + k_argmin, blocking copies): what the call cost before they existed
    import ctypes as C
    from stm32_speech_recognition_amd.engine import load_library
    L = load_library()
    L.sr_compat_engine.restype = C.c_void_p
    ce = C.c_void_p(L.sr_compat_engine())
    g_on = list(g)
    L.sr_set_small_launch(ce, C.c_int(1))
    out["spch_recg_batch_kernels_us"] = us(f_spch, n_utt)
    L.sr_set_small_launch(ce, C.c_int(0))
    out["spch_recg_small_launch_identical"] = bool(all(g_on[i] == g[i] for i in range(n_utt)))
    compat.spch_recg(pcm[0])
    if ref is not None:
        def f_rspch(i):
            r[i] = ref.spch_recg(pcm[i], store)
        out["spch_recg_reference_objects_us"] = us(f_rspch, n_utt)
        out["spch_recg_identical"] = bool(all(r[i][0] == 0 and g[i][1] == r[i][2] for i in range(n_utt)))
    # ---- get_mfcc on the VAD's first segment
    atap = compat.atap_tag()
    compat.noise_atap(pcm[0], 2400, atap)
    segs = compat.VAD(pcm[0], S, atap)
    s0, e0 = segs[0]
    ftrs = [None]

    def f_mfcc(i):
        ftrs[0] = compat.get_mfcc(pcm[0], s0, e0, atap)
    out["get_mfcc_us"] = us(f_mfcc, 32, "get_mfcc")
    if ref is not None:
        ra, rseg = ref.vad(pcm[0])
        out["get_mfcc_reference_objects_us"] = us(lambda i: ref.mfcc(pcm[0], int(rseg[0]), int(rseg[1]), ra), 32)
    # ---- dtw(): the slot scan of main.c:279-291
    slots = [compat.v_ftr_tag.from_buffer_copy(bytes(store[k * 4096:k * 4096 + 2860])) for k in range(Kl)]
    ins = []
    for i in range(8):
        compat.noise_atap(pcm[i], 2400, atap)
        sg = compat.VAD(pcm[i], S, atap)
        ins.append(compat.get_mfcc(pcm[i], sg[0][0], sg[0][1], atap))
    sc = np.zeros((8, Kl), np.uint32)

    def f_scan(i):
        for k in range(Kl):
            sc[i % 8, k] = compat.dtw(ins[i % 8], slots[k])
    out["dtw_slot_scan_us"] = us(f_scan, 16, "dtw_slot_scan")
    out["dtw_per_call_us"] = out["dtw_slot_scan_us"] / Kl
    if ref is not None:
        rslots = [np.frombuffer(bytes(store[k * 4096:k * 4096 + 2860]), np.uint8).copy() for k in range(Kl)]
        rins = [np.frombuffer(bytes(f), np.uint8).copy() for f in ins]
        rsc = np.zeros((8, Kl), np.uint32)

        def f_rscan(i):
            for k in range(Kl):
                rsc[i % 8, k] = ref.dtw(rins[i % 8], rslots[k])
        out["dtw_slot_scan_reference_objects_us"] = us(f_rscan, 16)
        out["dtw_identical"] = bool(np.array_equal(sc, rsc))
    # ---- batched device-resident entry point at small B
    dpcm = torch.from_numpy(pcm.view(np.int16)).to(dev)
    dpcm = dpcm.repeat((256 + n_utt - 1) // n_utt, 1)[:256].contiguous()
    for Bs in (1, 16, 256):
        o = eng.alloc_outputs(Bs, dev, mfcc=False, vad=False)

        def f_dev(i):
            eng.recognize_dev(dpcm[:Bs], o)
            torch.cuda.synchronize()
        t = us(f_dev, 50, f"sr_recognize_batch_dev_B{Bs}")
        out[f"sr_recognize_batch_dev_B{Bs}_us"] = t
        out[f"sr_recognize_batch_dev_B{Bs}_us_per_utterance"] = t / Bs
        # where the time goes: hipEvents around each of the call's four kernels (the events themselves add a few us)
        eng.set_profiling(True)
        for i in range(20):
            f_dev(i)
        sm = eng.stage_ms()
        eng.set_profiling(False)
        out[f"sr_recognize_batch_dev_B{Bs}_kernel_us"] = {k: sm[k] * 1e3 for k in ("vad", "mfcc", "dtw", "argmin", "total")}
        if Bs == 1:  # the batch kernels on the same call (small-launch mode 1)
            eng.set_small_launch(1)
            out["sr_recognize_batch_dev_B1_batch_kernels_us"] = us(f_dev, 50)
            eng.set_small_launch(0)
    out["p90_us"] = p90
    eng.close()
    return out


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when a communicator is created
    (five lines: "RCCL version : ...", "HIP version", "ROCm version", "Hostname", "Librccl path"), and other libraries may do
    the same: file descriptor 1 is pointed at stderr for the whole run and the JSON line goes out through a private
    duplicate of the original stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.launcher == "single":
            return run_single_process(args)
        return self_launch(args)
    return run_rank(args)


def run_rank(args):
    out_stream = claim_stdout()
    rank, local_rank, world = du.env_rank()
    # test hooks (tests/test_gpu_parity.py runs the N = 2 code path on a 1-GPU box): collective backend and a forced
    # device ordinal.  The driver never sets them: N > 1 means one rank per GPU over RCCL.
    backend = os.environ.get("SR_BENCH_BACKEND", "nccl")
    if "SR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SR_BENCH_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: device {local_rank} requested but {torch.cuda.device_count()} MI355X visible "
                         "(there is no CPU path; --gpus N needs N devices)")
    global FORCE_DIST
    FORCE_DIST = os.environ.get("SR_BENCH_FORCE_DIST") == "1"
    dist = None
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist
        if world == 1:  # test hook: a one-rank job still needs a rendezvous
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        du.init_process_group(backend, local_rank)
    torch.cuda.set_device(local_rank)
    B = args.batch
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"--scaling strong: the global batch {B} must divide over {world} GPUs")
        B = B // world  # this rank's shard of the fixed global batch
        args.batch_global, args.batch = args.batch, B
    if args.scorer == "dp":
        if world != 1 or args.workload != "ref":
            raise SystemExit("--scorer dp is a single-GPU side measurement of the reference workload")
        e = measure_dp(B, args.templates or K_DEFAULT, args.steps, args.dp_lanes, local_rank)
        line = {"metric": f"utterances/sec (256-frame, {args.templates or K_DEFAULT} templates; OPT-IN full-DP scorer alone, NON-REFERENCE, "
                          "not the headline metric)", "value": e["value"], "unit": "utterances/s", "n_gpus": 1, "steps": args.steps,
                "warmup": 1, "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "int32", "data": "synthetic", "config": {"workload": e["workload"]}, "dp": e}
        out_stream.write(json.dumps(line) + "\n")
        out_stream.flush()
        return 0
    m = measure(args.workload, B, args.templates, args.steps, args.warmup, rank, world, local_rank, dist, gain=args.gain,
                exchange=args.exchange)
    if rank == 0:
        line = headline(args, m, world, backend)
        if FORCE_DIST:
            line["config"]["parallelism"] += " -- TEST HOOK: process group and score exchange forced at N = 1"
        if world > 1 and not args.no_cpu_baseline:
            # N > 1: the line still carries the host baseline and a parity flag -- over a sample of EVERY rank's shard,
            # read from the gathered matrix (rank 0 cannot see the other ranks' buffers: they came in one extra all-gather)
            ns = min(PER_RANK_PARITY_N, B)
            sp = synth.as_u16_numpy(m["sample_pcm"])
            n0 = min(args.cpu_sample, 1024, B)
            hosts = [synth.as_u16_numpy(m["pcm"][:n0])] + [sp[r * ns:(r + 1) * ns] for r in range(1, world)]
            line["cpu_baseline"] = multi_rank_cpu_baseline(hosts, m["gathered"], B, m["tm"], m["tfr"], m["eng_cfg"], n0,
                                                           results_from_torch(m["out"]["results"][:n0]), kind=m["exchange_kind"],
                                                           own_scores=m["out"]["scores"][:n0])
            line["workload_stats"] = workload_stats(hosts[0][:WORKLOAD_STATS_N], m["eng_cfg"], args.gain)
        if world == 1 and not args.no_cpu_baseline:
            line["workload_stats"] = workload_stats(synth.as_u16_numpy(m["pcm"][:WORKLOAD_STATS_N]), m["eng_cfg"], args.gain)
            line["cpu_baseline"] = cpu_baseline(m["pcm"], m["eng"], m["out"], m["tm"], m["tfr"], args.cpu_sample, m["eng_cfg"],
                                                one_thread_n=64)
            if args.workload == "ref":
                line["cpu_reference_objects"] = cpu_reference_objects(local_rank)
                line["latency"] = latency_block(local_rank)
                # one capture at THIS workload's shapes (256 frames, the store of the timed steps, 320-frame cap), same engine
                eng, o1 = m["eng"], m["eng"].alloc_outputs(1, m["pcm"].device, mfcc=False, vad=False)
                x1 = m["pcm"][:1].contiguous()
                for mode, key in ((0, "one_capture_at_the_benchmark_shapes_us"), (1, "one_capture_at_the_benchmark_shapes_batch_kernels_us")):
                    eng.set_small_launch(mode)
                    ts = []
                    for i in range(24):
                        t0 = time.perf_counter()
                        eng.recognize_dev(x1, o1)
                        torch.cuda.synchronize()
                        ts.append((time.perf_counter() - t0) * 1e6)
                    line["latency"][key] = sorted(ts[4:])[10]
                eng.set_small_launch(0)
        default_shape = args.workload == "ref" and (B, m["K"]) == (65536, 100)
        if world == 1 and (default_shape or args.other_scale > 1) and not args.no_other_configs:
            # the other single-GPU shapes BASELINE.json names, under the same clock (never part of `value`)
            del m["pcm"], m["out"]
            m["eng"].close()
            torch.cuda.empty_cache()
            cpu_n = 0 if args.no_cpu_baseline else 256
            sc = args.other_scale
            line["other_configs"] = [other_config("ref", 4096 // sc, 10, args.other_steps, local_rank, cpu_n),
                                     other_config("ext", 65536 // sc, 500, args.other_steps, local_rank, cpu_n and 128)]
            if args.gain == 1.0:  # the metric's configuration once more, at the amplitudes SURVEY.md 8(d) specifies (200-600)
                line["other_configs"].insert(0, other_config("ref", 65536 // sc, 100, args.other_steps, local_rank, cpu_n, gain=SURVEY_GAIN))
            # the scorer BASELINE.json's north_star describes (anti-diagonal wavefront), opt-in and non-reference: timed alone
            line["other_configs"].append(measure_dp(65536 // sc, K_DEFAULT, min(3, args.other_steps), 0, local_rank,
                                                    parity_n=64 if cpu_n else 0))
        out_stream.write(json.dumps(line) + "\n")
        out_stream.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


_PMC_PATH = ("k_vad.hip", "sr_vad_dev.h", "k_mfcc.hip", "k_dtw.hip", "sr_dev.h", "sr_fft_dev.h", "sr_dtw_dev.h", "sr_device.h", "sr_tables.h")
PMC_SOURCES = {  # the translation units (+ the device headers they include) each committed PMC file depends on
    # (sr_tables.h holds the tier bounds and fused multipliers that steer k_mfcc's instruction mix)
    "pmc_traffic.json": ("k_mfcc.hip", "sr_dev.h", "sr_fft_dev.h", "sr_device.h", "sr_tables.h"),
    "pmc_valu.json": _PMC_PATH,
    "pmc_valu_loud.json": _PMC_PATH,
    "pmc_valu_dp.json": ("k_dtw_dp.hip", "sr_dev.h", "sr_dtw_dev.h", "sr_device.h", "sr_tables.h"),
    "pmc_valu_k10.json": _PMC_PATH,
    "pmc_valu_ext.json": ("k_vad.hip", "sr_vad_dev.h", "k_mfcc_ext.hip", "k_dtw.hip", "sr_dev.h", "sr_fft_dev.h", "sr_dtw_dev.h", "sr_device.h",
                          "sr_tables.h"),
}


def valu_pmc_file(workload, B, Kt, gain=1.0):
    """the committed PMC file (issue slots per utterance) that belongs to a workload shape and amplitude, or None"""
    if workload == "ext":
        return "pmc_valu_ext.json" if (Kt == 500 and gain == 1.0) else None
    if gain == SURVEY_GAIN:
        return "pmc_valu_loud.json" if Kt == 100 else None
    if gain != 1.0:
        return None
    return "pmc_valu.json" if Kt == 100 else "pmc_valu_k10.json" if Kt == 10 else None


def valu_roofline(which, B, step_ms, sclk):
    """What actually bounds the path: VALU issue.  Measured directly on this chip (profiles/r02/VALU_ISSUE.md,
    profiles/valu_issue_ubench.hip): in a mixed instruction stream every wave64 VALU instruction holds its SIMD's issue
    port for 4 cycles (a transcendental for 8); the 2-cycle rate of simple ops needs a pure run of them, which these
    kernels never have.  Ceiling = 1024 SIMDs x 2.4 GHz / 4 issue slots per second; the slots a step needs come from the
    committed PMC pass of the same workload shape (SQ_ACTIVE_INST_VALU = 4-cycle issue slots, profiles/pmc_valu*.json).
    Returns (dict or None, stale flag or None)."""
    path = os.path.join(ROOT, "profiles", which) if which else None
    if not path or not os.path.exists(path):
        return None, None
    try:
        vj = json.load(open(path))
        if not pmc_is_current(vj, which):
            return None, True
        insts = sum(v for k, v in vj.items() if k.endswith("_valu_insts_per_utt"))
        slots = sum(v for k, v in vj.items() if k.endswith("_valu_slots_per_utt")) or insts
        peak = 1024 * 2.4e9 / 4.0
        achv = slots * B / (step_ms * 1e-3)
        return {"bound": "valu-issue", "achieved": achv, "peak": peak, "unit": "4-cycle issue slots/s",
                "frac": achv / peak, "valu_slots_per_utt": slots, "valu_insts_per_utt": insts,
                "cycles_per_slot": 4.0, "clock_hz_assumed": 2.4e9,
                "clock_hz_measured": sclk["mean_mhz"] * 1e6 if sclk else None,
                "frac_at_measured_clock": achv / (1024 * sclk["mean_mhz"] * 1e6 / 4.0) if sclk else None,
                # the SECOND denominator: MI355X_MICROARCH.md's datasheet rate of one wave64 VALU instruction per SIMD per 2
                # cycles (reached only by a pure run of full-rate ops, profiles/r02/VALU_ISSUE.md) -- instructions, not slots
                "frac_vs_2cycle_peak": insts * B / (step_ms * 1e-3) / (1024 * 2.4e9 / 2.0),
                "peak_2cycle": 1024 * 2.4e9 / 2.0,
                "source": vj.get("source"), "pmc_file": "profiles/" + which,
                "rates_source": "profiles/r02/VALU_ISSUE.md (per-opcode s_memtime micro-benchmark)",
                "note": "derived: slot counts from the committed PMC pass x this run's step time; the chip "
                        "clocks 2.3-2.4 GHz under this load, the ceiling assumes the nominal 2.4"}, False
    except Exception:
        return None, None


def kernel_sources_sha(which):
    """sha256 over the kernel sources a committed PMC file (profiles/pmc_*.json) was measured with; profiles/summarize.py
    stores it in the file, and a figure whose kernels have changed since is reported as stale instead of being used"""
    import hashlib
    d = os.path.join(ROOT, "stm32_speech_recognition_amd", "csrc")
    h = hashlib.sha256()
    for f in PMC_SOURCES[which]:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_is_current(j, which):
    return j.get("kernel_sources_sha") == kernel_sources_sha(which)


def headline(args, m, world, backend="nccl", launcher=None):
    """the ONE JSON line of the contract, from a measure() result"""
    B, Kt, S, rate = args.batch, m["K"], m["S"], m["rate"]
    stage, stage_iso, dt = m["stage"], m["stage_iso"], m["dt"]
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    C = 12
    by_path = algorithmic_bytes_per_utt(S, T, C, Kt)
    by_mfcc = mfcc_bytes(rate)
    # the engine cuts a step into chunks on overlapping streams: each kernel is launched `launches` times per step,
    # one launch covers B / launches utterances; stage[...] are per-launch durations (hipEvents on its stream)
    launches = stage["launches_per_call"]
    ach_ovl = by_mfcc * (B / launches) / (stage["mfcc"] * 1e-3) / 1e9
    # the same kernel alone on the chip: ONE launch over all B utterances (the extra pass right after the timed steps)
    ach_iso = by_mfcc * B / (stage_iso["mfcc"] * 1e-3) / 1e9
    traffic = traffic_iso = traffic_src = None
    traffic_stale = valu_stale = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if args.workload == "ref" and os.path.exists(tpath):
        try:  # the PMC passes measured one whole-batch launch over tj["B"] utterances; traffic scales with utterances
            tj = json.load(open(tpath))
            traffic_stale = not pmc_is_current(tj, "pmc_traffic.json")
            if not traffic_stale:
                traffic = tj["k_mfcc_hbm_bytes_per_launch"] * (B / launches) / tj["B"]
                traffic_iso = tj["k_mfcc_hbm_bytes_per_launch"] * B / tj["B"]
            traffic_src = tj.get("source")
        except Exception:
            traffic = traffic_iso = None
    gain = m.get("gain", 1.0)
    roofline_valu, valu_stale = valu_roofline(valu_pmc_file(args.workload, B, Kt, gain), B, stage["total"], m.get("sclk"))
    par = f"utterance-sharded x{world}"
    if world > 1:
        what = "scores" if m.get("exchange_kind", "scores") == "scores" else "result records (--exchange results)"
        par += f", RCCL all-gather of {what}" if backend == "nccl" else f", all-gather of {what} over {backend} (TEST HOOK, not RCCL)"
    line = {
        "metric": f"utterances/sec (256-frame, {Kt} templates)" if args.workload == "ref"
        else f"utterances/sec (EXTENSION: 16 kHz/512-pt/40 Mel, 256-frame, {Kt} templates; no reference counterpart)",
        "value": value,
        "unit": "utterances/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": getattr(args, "scaling", "weak"),
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "exchange": m.get("exchange"),
        "config": {"workload": workload_name(args.workload, B, Kt, gain) + (f" (strong scaling: global batch {args.batch_global})"
                                                                             if getattr(args, "batch_global", None) else ""),
                   "batch_per_gpu": B, "templates": Kt, "frames": T, "buf_len": S, "parallelism": par, "gain": gain},
        "workload_stats": None,  # filled by the CPU legs (see workload_stats)
        # HBM roofline of the dominant kernel, as the contract defines it: algorithmic bytes of one launch / the duration
        # of that launch.  achieved / frac = the kernel ALONE on the chip, one launch over the whole batch (hipEvents on
        # its stream, in the pass right after the timed steps) -- the figure profiles/*_rocprof_summary.csv reproduces.
        "roofline": {"bound": "hbm", "kernel": "k_mfcc" if rate == 1 else "k_mfcc_ext", "achieved": ach_iso,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_iso / HBM_PEAK_GBS, "traffic": traffic_iso,
                     "traffic_source": traffic_src, "traffic_stale": traffic_stale, "algorithmic_bytes_per_launch": by_mfcc * B,
                     "kernel_ms": stage_iso["mfcc"], "launches_per_step": 1, "utterances_per_launch": B,
                     "pass": "isolated (untimed pass right after the timed steps); `value` / `ms_per_step` come from the pipelined "
                             "timed steps, whose per-launch figures are under `overlapped`",
                     "measured": "hipEvents around the launch on its own stream; whole batch as ONE launch, kernel alone on "
                                 "the chip (extra pass right after the timed steps; agrees with the rocprof 'whole batch' row)",
                     "overlapped": {"achieved": ach_ovl, "frac": ach_ovl / HBM_PEAK_GBS, "kernel_ms": stage["mfcc"],
                                    "launches_per_step": launches, "utterances_per_launch": B / launches,
                                    "algorithmic_bytes_per_launch": by_mfcc * B / launches, "traffic": traffic,
                                    "note": "launches of the TIMED steps: each covers B/launches utterances and shares the "
                                            "chip with two other chunks' kernels on other streams, so the per-launch "
                                            "durations overlap and do not add up to the step; not a fraction of the chip"},
                     # the fractions that describe the TIMED steps, inside this object (the driver's record keeps `roofline`):
                     "timed_step_frac": by_path * B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "timed_step_note": "SURVEY 8(d)'s whole-path algorithmic bytes per utterance x B / ms_per_step / 8 TB/s",
                     "valu_frac": roofline_valu["frac"] if roofline_valu else None,
                     "valu_frac_at_measured_clock": roofline_valu["frac_at_measured_clock"] if roofline_valu else None,
                     "valu_frac_vs_2cycle_peak": roofline_valu["frac_vs_2cycle_peak"] if roofline_valu else None,
                     "valu_note": "the BINDING ceiling: 4-cycle VALU issue slots of a timed step / (1024 SIMDs x clock / 4), "
                                  "details under roofline_valu; null when the committed PMC file is stale or absent for this shape",
                     "note": "the path is integer-VALU-issue-bound, not HBM-bound (DESIGN.md 3.2): see roofline_valu"},
        "roofline_path": {"bytes_per_utt": by_path, "achieved": by_path * B / (stage["total"] * 1e-3) / 1e9,
                          "unit": "GB/s", "frac": by_path * B / (stage["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "note": "whole step (all chunks, fork -> join on the launch stream)"},
        "roofline_valu": roofline_valu,
        "roofline_valu_stale": valu_stale,  # true: profiles/pmc_valu.json was measured with other kernel sources -> figure dropped
        "kernel_ms": stage,
        "kernel_ms_isolated": {**stage_iso, "note": "untimed extra pass: whole batch as one chunk on one stream (no overlap)"},
        "top1_word_accuracy": m["acc"],
        "sclk": m.get("sclk"),
    }
    if launcher:
        line["launcher"] = launcher
    return line


def run_single_process(args):
    """--launcher single: ONE process drives the N GPUs through the C ABI (sr_multi_*, csrc/sr_multi.cpp): one engine
    and one stream per device, device-resident shards, and per step one grouped in-place ncclAllGather of the score
    matrix on the same streams (the exchange step of main.c:279-291's slot scan, for every utterance, on every GPU)."""
    from stm32_speech_recognition_amd.engine import MultiEngine
    out_stream = claim_stdout()
    n, B = args.gpus, args.batch
    if args.workload != "ref":
        raise SystemExit("--launcher single runs the reference workload only")
    devs = list(range(n))
    testing = False
    if "SR_BENCH_DEVICE" in os.environ:  # test hook: all "ranks" on one device (needs the fake RCCL named by SR_RCCL_LIBRARY
        devs = [int(os.environ["SR_BENCH_DEVICE"])] * n  # and the -DSR_TESTING build of the library, which has multi_allow_dup)
        from stm32_speech_recognition_amd.engine import dev_hook
        dev_hook("multi_allow_dup", 1)
        testing = True
    if not torch.cuda.is_available() or max(devs) >= torch.cuda.device_count():
        raise SystemExit(f"--gpus {n} but {torch.cuda.device_count()} MI355X visible (there is no CPU path)")
    rate, eng_cfg, Kt, n_words = workload_setup("ref", args.templates)
    S = synth.buf_len_for(T, rate)
    me = MultiEngine(devs, max_frames=MAX_FRAMES, testing=testing)
    bank = synth.word_bank(n_words)
    e0 = me.engine(0)
    tm, tfr, rng = make_templates(e0, bank, Kt, n_words, rate, torch.device("cuda", devs[0]))
    me.set_templates_dense(tm, tfr.astype(np.uint32))
    words = torch.from_numpy(rng.integers(0, n_words, n * B))
    pl, rl, al, sl = [], [], [], []
    for i, d in enumerate(devs):
        dev = torch.device("cuda", d)
        pl.append(synth.make_utterances(words[i * B:(i + 1) * B], [T] * B, seed=1000 + i, bank=bank, S=S, device=dev, rate=rate))
        rl.append(torch.empty(B, 4, dtype=torch.int32, device=dev))
        al.append(torch.empty(n * B, Kt, dtype=torch.int32, device=dev))
        sl.append(torch.cuda.Stream(device=dev))

    def sync():
        for d in set(devs):
            torch.cuda.synchronize(d)

    sync()
    for _ in range(args.warmup):
        me.recognize_dev(pl, rl, al, streams=sl)
    sync()
    e0.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        me.recognize_dev(pl, rl, al, streams=sl)
    sync()
    dt = time.perf_counter() - t0
    stage = e0.stage_ms()
    e0.set_profiling(False)
    e0.set_pipeline(streams=1)
    e0.set_profiling(True)
    out0 = e0.alloc_outputs(B, torch.device("cuda", devs[0]), mfcc=False, vad=False)
    for _ in range(2):
        e0.recognize_dev(pl[0], out0, stream=sl[0].cuda_stream)
    sync()
    stage_iso = e0.stage_ms()
    e0.set_profiling(False)
    e0.set_pipeline()
    # every device holds the whole matrix; its own block is what its kernels wrote; results are 256-frame matches
    ref_all = al[0].cpu()
    accs = []
    for i in range(n):
        assert torch.equal(al[i].cpu(), ref_all), f"device {i}: gathered score matrix differs from device 0's"
        res = results_from_torch(rl[i])
        assert (res["status"] == 0).all() and (res["frm_num"] == T).all(), "workload is not 256-frame utterances"
        accs.append((res["best_tpl"] % n_words == words[i * B:(i + 1) * B].numpy()))
    assert torch.equal(ref_all[:B], out0["scores"].cpu()), "device 0's block differs from its single-engine scores"
    m = dict(dt=dt, stage=stage, stage_iso=stage_iso, acc=float(np.concatenate(accs).mean()), S=S, rate=rate, K=Kt)
    line = headline(args, m, n, launcher="single (one process, sr_multi_* C ABI: one engine + stream per device, grouped "
                                         "in-place ncclAllGather per step)")
    line["config"]["parallelism"] = f"utterance-sharded x{n}, RCCL all-gather of scores (single process, sr_multi)"
    if os.environ.get("SR_RCCL_LIBRARY"):
        line["config"]["parallelism"] += " -- TEST HOOK: collective library " + os.environ["SR_RCCL_LIBRARY"]
    line["exchange"] = {"backend": "rccl (sr_multi: grouped in-place ncclAllGather)", "ranks_in_communicator": n,
                        "allgather_bytes_per_rank_out": int(n * B * Kt * 4),
                        "every_device_holds_identical_gathered_matrix": True}  # asserted above
    if not args.no_cpu_baseline:
        ns, n0 = min(PER_RANK_PARITY_N, B), min(args.cpu_sample, 1024, B)
        hosts = [synth.as_u16_numpy(pl[0][:n0])] + [synth.as_u16_numpy(pl[i][:ns]) for i in range(1, n)]
        line["cpu_baseline"] = multi_rank_cpu_baseline(hosts, al[0], B, tm, tfr, eng_cfg, n0, results_from_torch(rl[0][:n0]))
    out_stream.write(json.dumps(line) + "\n")
    out_stream.flush()
    me.close()
    return 0


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


def cpu_baseline(pcm, eng, out, tm, tfr, n, eng_cfg, one_thread_n=0):
    """The reference C path on the host cores, on the first n utterances of the timed batch (see cpu_baseline_arrays)."""
    host = synth.as_u16_numpy(pcm[:n])
    gsc = out["scores"][:n].cpu().numpy().view(np.uint32)
    gres = results_from_torch(out["results"][:n])
    return cpu_baseline_arrays(host, gsc, gres["best_tpl"], gres["min_dis"], tm, tfr, eng_cfg,
                               f"first {n} utterances of the timed batch", one_thread_n)


PER_RANK_PARITY_N = 128  # N > 1: utterances of EVERY rank's shard that are checked against the CPU reference


def multi_rank_cpu_baseline(host_per_rank, gathered, B, tm, tfr, eng_cfg, n0, own_results=None, kind="scores", own_scores=None):
    """N > 1 (rank 0 / the single process): `cpu_baseline` + a parity flag that covers every rank's shard.  The sample is the
    first n0 utterances of rank 0's shard followed by the first len(host_per_rank[r]) utterances of every other rank's
    shard; the GPU side of the comparison is read from the GATHERED score matrix (what the exchange step delivered:
    rank r's block starts at row r * B), with argmin / min distance re-derived from it by the strict-< first-minimum
    scan of main.c:279-291 (dist_util.argmin_first) -- for rank 0 they are also compared with the kernel's own result
    records.  One pass of the reference's objects over the sample gives the timing and the parity."""
    world = len(host_per_rank)
    rows, hosts, owner = [], [], []
    for r in range(world):
        n_r = min(n0 if r == 0 else len(host_per_rank[r]), len(host_per_rank[r]), B)
        rows.append(np.arange(r * B, r * B + n_r))
        hosts.append(host_per_rank[r][:n_r])
        owner += [r] * n_r
    rows = np.concatenate(rows)
    host = np.concatenate(hosts)
    g = gathered[torch.from_numpy(rows).to(gathered.device)]
    if kind == "results":
        # --exchange results: the gathered rows ARE the kernels' result records (argmin and distance of every rank's utterances);
        # the score matrix stays on its rank, so scores are compared for rank 0's rows only (cpu_baseline_arrays: first len(gsc) rows)
        rec = results_from_torch(g)
        gbest, gdis = rec["best_tpl"].astype(np.uint32), rec["min_dis"].astype(np.uint32)
        gsc = own_scores[:len(hosts[0])].cpu().numpy().view(np.uint32)
    else:
        best, mn = du.argmin_first(g)
        gsc = g.cpu().numpy().view(np.uint32)
        gbest, gdis = best.cpu().numpy().astype(np.uint32), mn.cpu().numpy().astype(np.uint32)
    own_ok = None
    if own_results is not None:  # rank 0's kernels' own records against the scan of the gathered rows
        k = min(len(own_results), len(rows), n0)
        own_ok = bool(np.array_equal(own_results["best_tpl"][:k], gbest[:k]) and np.array_equal(own_results["min_dis"][:k], gdis[:k]))
    cb = cpu_baseline_arrays(host, gsc, gbest, gdis, tm, tfr, eng_cfg,
                             f"first {len(hosts[0])} utterances of rank 0's shard + first {PER_RANK_PARITY_N} of each of the other "
                             f"{world - 1} ranks' shards ({len(rows)} utterances; GPU " +
                             ("scores taken from the gathered matrix)" if kind == "scores" else
                              "argmin / distance taken from the gathered result records, scores from rank 0's own matrix)"),
                             one_thread_n=32)
    cb["per_rank_sample"] = {"ranks": world, "utterances_per_rank": [int(len(h)) for h in hosts],
                             "gpu_side": ("rows r*B .. of the all-gathered score matrix on rank 0; argmin / min_dis re-derived by the "
                                          "strict-< slot scan (main.c:279-291)") if kind == "scores" else
                                         "rows r*B .. of the all-gathered result records on rank 0",
                             "rank0_result_records_agree_with_gathered_scan": own_ok}
    if own_ok is False:
        cb["gpu_results_identical_on_sample"] = False
    return cb


def cpu_baseline_arrays(host, gsc, gbest, gdis, tm, tfr, eng_cfg, what, one_thread_n=0):
    """The reference C path on the host cores over the capture buffers `host` (u16 [n, S]), cross-checked against the GPU's
    scores `gsc` (u32 [n, K]), argmin `gbest` and distance `gdis` of the same utterances.

    kind "reference" (reference workload): the reference's OWN VAD.C / MFCC.C / DTW.C objects (oracle/_ref/libsr_ref320.so:
    compiled from /root/reference where the sources lie, with the one compile-time constant that caps a record at 119
    frames raised to 320, oracle/Makefile) + the C transcription of the assembly FFT + the restated spch_recg scan.  The
    objects keep file-scope statics (MFCC.C:14-15, DTW.C:65-68), so each host thread dlopens its own private copy of the
    .so; ctypes drops the GIL during the calls.  kind "port": the parametrised restatement (oracle tier ii), used for the
    extension workload (no reference counterpart) or when the reference objects are absent; it is also timed as a side
    figure.  one_thread_n > 0: the first one_thread_n utterances are also timed on ONE host thread (`cores_1`,
    SURVEY.md 8(d): "1 core, then all cores")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cores, hw_threads, quota = usable_cores()
    n = host.shape[0]
    gres = {"best_tpl": np.asarray(gbest), "min_dis": np.asarray(gdis)}
    where = (f"{what}, {cores} host threads (box: {hw_threads} hardware threads, "
             f"cgroup CPU quota {quota if quota else 'none'})")
    # ---- port (tier ii), multi-threaded inside the C library
    orc = ol.Oracle(max_frames=MAX_FRAMES, **eng_cfg)
    tpl = orc.make_templates(tm, tfr.astype(np.uint32))
    orc.recognize_batch(host[:cores * 2], tpl, n_threads=cores, want_mfcc=False, want_scores=False)  # warm-up
    t0 = time.perf_counter()
    ores, _, osc = orc.recognize_batch(host, tpl, n_threads=cores, want_mfcc=False, want_scores=True)
    dt_port = time.perf_counter() - t0
    ns = len(gsc)  # scores may cover only the leading rows (--exchange results: the other ranks' matrices are not gathered)
    match_port = bool(np.array_equal(gsc, osc[:ns]) and np.array_equal(gres["best_tpl"], ores["best_tpl"]))
    port = {"value": n / dt_port, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": where + ", gcc -O2 oracle (tier ii)", "seconds": dt_port,
            "gpu_results_identical_on_sample": match_port}
    n1 = min(one_thread_n, n)
    if n1:
        t0 = time.perf_counter()
        orc.recognize_batch(host[:n1], tpl, n_threads=1, want_mfcc=False, want_scores=False)
        port["cores_1"] = {"value": n1 / (time.perf_counter() - t0), "unit": "utterances/s", "cores": 1, "utterances": n1}
    if eng_cfg or not ol.RefLib320.available():
        return port
    # ---- the reference's own objects, one private copy of the .so per thread (tests/oracle_lib.py Ref320Pool)
    Kt = len(tfr)
    stride = 8192
    store = ol.ref320_store(tm, tfr, None, stride)
    pool = ol.Ref320Pool(cores)
    pool.recognize(host, store, Kt, stride, n_run=min(n, 2 * cores))   # warm-up
    rr = pool.recognize(host, store, Kt, stride)
    dt = rr["seconds"]
    r_st, r_sc, r_best, r_dis = rr["status"], rr["scores"], rr["best"], rr["dis"]
    cores_1 = None
    if n1:                                                  # the same objects on ONE host thread
        r1 = pool.recognize(host, store, Kt, stride, n_run=n1, threads=1)
        cores_1 = {"value": n1 / r1["seconds"], "unit": "utterances/s", "cores": 1, "utterances": n1}
    pool.close()
    match = bool((r_st == 0).all() and np.array_equal(gsc, r_sc[:ns]) and np.array_equal(gres["best_tpl"], r_best)
                 and np.array_equal(gres["min_dis"], r_dis))
    return {"value": n / dt, "unit": "utterances/s", "cores": cores, "kind": "reference",
            "sample": where + ", the reference's own VAD.C/MFCC.C/DTW.C objects (gcc -O2, vv_tim_max raised to 320 frames) "
                              "+ C transcription of the asm FFT, one private .so copy per thread",
            "seconds": dt, "gpu_results_identical_on_sample": match, "cores_1": cores_1, "port": port}


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
    bank = synth.word_bank(N_WORDS_DEFAULT)
    rng = np.random.default_rng(7)
    tfr = rng.integers(60, T + 1, Kr)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(Kr) % N_WORDS_DEFAULT, tfr, seed=5, bank=bank, S=S))
    store, st = eng.train_store(tp, np.arange(Kr), n_slots=Kr)       # save_mdl flash image (main.c:121-138)
    assert (st == 0).all()
    eng.set_templates_store(store)
    words = rng.integers(0, N_WORDS_DEFAULT, n)
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
    sys.exit(main())
