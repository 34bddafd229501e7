"""Workgroups of the frame kernel (development hook mfcc_grid, -DSR_TESTING build): step time and the kernel alone for
    python profiles/experiments/mfcc_grid_sweep.py ref|ext [grid,grid,...]
on the benchmark's workload.  0 = the library's own choice (4 x the resident workgroups)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd import engine as E


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "ref"
    grids = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2048, 4096, 8192, 16384, 32768, 0]
    B = 65536
    rate, cfg, Kt, n_words = bench.workload_setup(wl, None)
    dev = torch.device("cuda", 0)
    rows = []
    pcm = None
    for gv in grids:
        E.dev_hook("mfcc_grid", gv)
        eng = Engine(max_frames=bench.MAX_FRAMES, device=0, testing=True, **cfg)
        bank = synth.word_bank(n_words)
        tm, tfr, rng = bench.make_templates(eng, bank, Kt, n_words, rate, dev)
        eng.set_templates_dense(tm, tfr.astype(np.uint32))
        if pcm is None:
            pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, n_words, B)), [bench.T] * B, seed=1000, bank=bank,
                                        S=synth.buf_len_for(bench.T, rate), device=dev, rate=rate)
        out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
        for _ in range(3):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        eng.set_pipeline(streams=1)
        eng.set_profiling(True)
        for _ in range(3):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        sm = eng.stage_ms()
        eng.set_profiling(False)
        rows.append({"grid": gv, "step_ms": round(ms, 3), "mfcc_alone_ms": round(sm["mfcc"], 3)})
        print(wl, rows[-1], flush=True)
        eng.close()
    E.dev_hook("mfcc_grid", 0)
    print(json.dumps({wl: rows}))


main()
