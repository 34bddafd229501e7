"""Pipeline sweep at run time (sr_set_pipeline) for one workload: python profiles/experiments/pipe_tune_rt.py ref|ext [B] [SxC,SxC,...]
Prints ms per step for (streams, max_chunks) combinations; development aid, results in RESULTS.md."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "ext"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    rate, cfg, Kt, n_words = bench.workload_setup(wl, None)
    dev = torch.device("cuda", 0)
    eng = Engine(max_frames=bench.MAX_FRAMES, device=0, **cfg)
    bank = synth.word_bank(n_words)
    tm, tfr, rng = bench.make_templates(eng, bank, Kt, n_words, rate, dev)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, n_words, B)), [bench.T] * B, seed=1000, bank=bank,
                                S=synth.buf_len_for(bench.T, rate), device=dev, rate=rate)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    res = {}
    combos = [(1, 1), (2, 8), (2, 12), (3, 6), (3, 9), (3, 12), (3, 15), (3, 18), (3, 24), (4, 8), (4, 12), (4, 16), (4, 24), (2, 16), (3, 12)]
    if len(sys.argv) > 3:
        combos = [tuple(int(v) for v in c.split("x")) for c in sys.argv[3].split(",")]
    for st, mc in combos:
        eng.set_pipeline(streams=st, min_chunk=max(1, B // 64), max_chunks=mc)
        for _ in range(2):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        res[f"{st}x{mc}"] = round(ms, 2)
        print(wl, "streams", st, "max_chunks", mc, "ms/step", round(ms, 2), flush=True)
    print(json.dumps({wl: res}))

main()
