"""configs[4] (16 kHz front end, 500 templates) as a pipelined step for forced DTW workgroup shapes (development hooks):
does a smaller DTW workgroup, which leaves LDS for a co-resident frame-kernel workgroup, beat the isolated optimum 7 x 125?
    python profiles/experiments/ext_dtw_u_sweep.py"""
import json, os, sys, time
os.environ.setdefault("SR_ENGINE_TESTING", "1")  # the development hooks exist only in the -DSR_TESTING build of the library
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd.engine import dev_hook

B = 65536
rate, cfg, Kt, n_words = bench.workload_setup("ext", None)
dev = torch.device("cuda", 0)
bank = synth.word_bank(n_words)
res = {}
pcm = None
for U, kc in ((0, 0), (4, 125), (5, 125), (6, 125), (8, 125), (5, 100), (6, 167), (3, 250), (4, 250)):
    dev_hook("dtw_u", U)
    dev_hook("dtw_kc", kc)
    eng = Engine(max_frames=bench.MAX_FRAMES, device=0, **cfg)
    tm, tfr, rng = bench.make_templates(eng, bank, Kt, n_words, rate, dev)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    if pcm is None:
        pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, n_words, B)), [bench.T] * B, seed=1000, bank=bank,
                                    S=synth.buf_len_for(bench.T, rate), device=dev, rate=rate)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    row = {}
    for st, mc in ((1, 1), (3, 6), (3, 12)):
        eng.set_pipeline(streams=st, min_chunk=1024, max_chunks=mc)
        for _ in range(2):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        row[f"{st}x{mc}"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
    res[f"U={U or 'auto'},Kc={kc or 'auto'}"] = row
    print(U, kc, row, flush=True)
    eng.close()
    del out
dev_hook("dtw_u", 0)
dev_hook("dtw_kc", 0)
print(json.dumps(res))
