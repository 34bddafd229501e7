"""Same-box A/B of builds of libsr_engine.so (the method behind every accept / reject in RESULTS.md).

    profiles/experiments/ab_build.sh base HEAD~1 ; profiles/experiments/ab_build.sh new .      (here: no GPU needed)
    gpurun -- 'python profiles/experiments/ab.py ab_libs/base.so ab_libs/new.so [rounds] [--gain G] [--workload ref|ext]
                                                 [--templates K] [--batch B]'

Every round runs each library once in a child process of its own (SR_ENGINE_LIB selects the library the package loads),
alternating, so that clock drift of the box hits all builds alike.  Per child: one line of JSON with
  pipe        ms per step of the chunked 3-stream pipeline (what bench.py times), 6 steps after one warm-up
  iso         ms per step as ONE chunk on one stream, and the hipEvent time of each kernel in that pass (vad / mfcc / dtw)
  chk         32-bit checksum of all scores (builds of one workload must agree: the arithmetic is bit-exact)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np


def child(a):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from stm32_speech_recognition_amd import Engine, synth
    ext = a.workload == "ext"
    rate, cfg = (2, dict(fs=16000, nfft=512, n_mel=40)) if ext else (1, {})
    T, K, B = 256, a.templates or (500 if ext else 100), a.batch
    NW = min(100 if ext else 25, K)
    dev = torch.device("cuda", 0)
    eng = Engine(max_frames=320, device=0, **cfg)
    bank = synth.word_bank(NW)
    rng = np.random.default_rng(2026)
    tfr = rng.integers(192, 321, K)
    tp = synth.make_utterances(np.arange(K) % NW, tfr, seed=77, bank=bank, S=synth.buf_len_for(320, rate), device=dev, rate=rate,
                               gain=a.gain)
    tvad, tmf = eng.features_dev(tp)
    torch.cuda.synchronize()
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    pcm = synth.make_utterances(rng.integers(0, NW, B), [T] * B, seed=1000, bank=bank, S=synth.buf_len_for(T, rate), device=dev,
                                rate=rate, gain=a.gain)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    res = {"gain": a.gain, "workload": a.workload, "K": K, "B": B}
    for name, streams in (("pipe", 3), ("iso", 1)):
        eng.set_pipeline(streams=streams) if streams == 1 else eng.set_pipeline()
        eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
        st = eng.stage_ms()
        eng.set_profiling(False)
        if streams == 1:
            res.update({k: round(st[k], 3) for k in ("vad", "mfcc", "dtw")})
    res["chk"] = int(out["scores"].to(torch.int64).sum().item() & 0xFFFFFFFF)
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--gain", type=float, default=1.0, help="speech amplitude factor of synth.make_utterances (SURVEY 8(d)'s amplitudes = 2.4)")
    ap.add_argument("--workload", choices=["ref", "ext"], default="ref")
    ap.add_argument("--templates", type=int, default=0)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    libs = [l for l in a.libs if not l.isdigit()]
    rounds = next((int(l) for l in a.libs if l.isdigit()), a.rounds)
    passthru = ["--gain", str(a.gain), "--workload", a.workload, "--templates", str(a.templates), "--batch", str(a.batch),
                "--steps", str(a.steps)]
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, SR_ENGINE_LIB=os.path.abspath(lib))
            p = subprocess.run([sys.executable, __file__, "x", "--child"] + passthru, env=env, capture_output=True, text=True)
            print(os.path.basename(lib), p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-600:], flush=True)


if __name__ == "__main__":
    main()
