"""Per-call cost of sr_recognize_batch_dev from 1 to 4096 captures at the firmware's shapes (16 000-sample capture, 110-frame
word, 80 slots of 70-119 frames), with the small-launch kernel forms switched off (sr_set_small_launch 1), automatic (0),
one workgroup per pair forced (2) and four lanes per pair forced (3, round 5): where each form stops paying.  Medians of the host wall clock over 30 calls + hipEvent-bracketed kernels.
    python profiles/experiments/small_launch_sweep.py > profiles/r05_small_launch_sweep.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stm32_speech_recognition_amd import Engine, synth  # noqa: E402


def main():
    # default: the firmware's shapes; `bench`: the benchmark's (256-frame captures, 100 templates of 192-320 frames, 320-frame cap)
    bench_shape = len(sys.argv) > 1 and sys.argv[1] == "bench"
    Tl, S, Kl, cap = (256, synth.buf_len_for(320), 100, 320) if bench_shape else (110, 16000, 80, 119)
    eng = Engine(max_frames=cap, device=0)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(4)
    tfr = rng.integers(192, 321, Kl) if bench_shape else rng.integers(70, 120, Kl)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(Kl) % 25, tfr, seed=8, bank=bank, S=S))
    stride = 4 + 24 * (cap + 1) if bench_shape else 4096
    store, st = eng.train_store(tp, np.arange(Kl), n_slots=Kl, stride=stride)
    assert (st == 0).all()
    eng.set_templates_store(store, stride=stride)
    n = 1024 if bench_shape else 4096
    dpcm = synth.make_utterances(rng.integers(0, 25, n), [Tl] * n, seed=9, bank=bank, S=S, device=torch.device("cuda", 0))
    rows = []
    ref = None
    for Bs in ((1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024) if bench_shape else (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)):
        o = eng.alloc_outputs(Bs, "cuda:0", mfcc=False, vad=False)
        row = {"B": Bs, "pairs": Bs * Kl}
        for mode, name in ((1, "batch_kernels"), (0, "automatic"), (2, "forced"), (3, "four_lanes_per_pair")):
            if mode == 2 and Bs > (16 if bench_shape else 256):
                continue  # one workgroup per pair at tens of thousands of pairs: milliseconds, nothing to learn
            eng.set_small_launch(mode)
            for _ in range(3):
                eng.recognize_dev(dpcm[:Bs], o)
                torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                eng.recognize_dev(dpcm[:Bs], o)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res = o["results"].cpu().numpy().copy()
            if mode == 1:
                ref = res
            eng.set_profiling(True)
            for _ in range(10):
                eng.recognize_dev(dpcm[:Bs], o)
                torch.cuda.synchronize()
            sm = eng.stage_ms()
            eng.set_profiling(False)
            row[name] = {"call_us": round(float(np.median(ts)) * 1e6, 1), "us_per_capture": round(float(np.median(ts)) * 1e6 / Bs, 2),
                         "kernel_us": {k: round(sm[k] * 1e3, 1) for k in ("vad", "mfcc", "dtw", "argmin")},
                         "identical_to_batch_kernels": bool(np.array_equal(res, ref))}
        rows.append(row)
    print(json.dumps({"shape": f"{S}-sample captures, {Tl}-frame words, {Kl} slots, {cap}-frame cap", "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
