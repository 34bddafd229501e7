"""Does a k_dtw_lds workgroup small enough to share a CU with frame-kernel workgroups (<= 40 KB of LDS: U = 3 at K = 100) make the
PIPELINED step faster, although the kernel alone is slower?  Development hooks, -DSR_TESTING library.
    python profiles/experiments/dtw_u_pipe.py [ref|ext]"""
import json, os, sys, time
os.environ.setdefault("SR_ENGINE_TESTING", "1")
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd.engine import dev_hook

which = sys.argv[1] if len(sys.argv) > 1 else "ref"
B = 65536
rate, cfg, Kt, n_words = bench.workload_setup(which, None)
dev = torch.device("cuda", 0)
bank = synth.word_bank(n_words)
shapes = {"ref": [(0, 0, 0), (3, 100, 8192), (4, 100, 8192), (0, 0, 0), (3, 100, 8192), (2, 100, 8192)],
          "ext": [(0, 0, 0), (3, 167, 8192), (4, 125, 8192), (0, 0, 0), (3, 125, 8192)]}[which]
pcm = None
for U, kc, g in shapes:
    dev_hook("dtw_u", U); dev_hook("dtw_kc", kc); dev_hook("dtw_tie_g", g)
    eng = Engine(max_frames=bench.MAX_FRAMES, device=0, **cfg)
    tm, tfr, rng = bench.make_templates(eng, bank, Kt, n_words, rate, dev)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    if pcm is None:
        pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, n_words, B)), [bench.T] * B, seed=1000, bank=bank,
                                    S=synth.buf_len_for(bench.T, rate), device=dev, rate=rate)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    row = {"U": U, "Kc": kc, "tie_g": g}
    for st, mc in ((1, 1), (2, 4), (3, 6), (3, 12), (3, 24), (4, 16)):
        eng.set_pipeline(streams=st, min_chunk=1024, max_chunks=mc)
        for _ in range(2):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        row[f"{st}x{mc}"] = round((time.perf_counter() - t0) / 6 * 1e3, 2)
    row["chk"] = int(out["scores"].to(torch.int64).sum().item() & 0xFFFFFFFF)
    print(json.dumps(row), flush=True)
    eng.close()
    del out
for h in ("dtw_u", "dtw_kc", "dtw_tie_g"):
    dev_hook(h, 0)
