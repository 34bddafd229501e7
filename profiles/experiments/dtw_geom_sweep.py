"""k_dtw_lds alone for forced workgroup shapes (development hooks of the -DSR_TESTING library): U utterances x Kc templates
per workgroup and the staged tie-table size, at the headline shape (K = 100) and configs[4]'s (K = 500).
    python profiles/experiments/dtw_geom_sweep.py [ref|ext]
Prints per shape: ms of the DTW kernel for the whole batch as one launch (hipEvents), and the 32-bit score checksum."""
import json, os, sys
os.environ.setdefault("SR_ENGINE_TESTING", "1")
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd.engine import dev_hook

which = sys.argv[1] if len(sys.argv) > 1 else "ref"
B = 65536
rate, cfg, Kt, n_words = bench.workload_setup(which if which == "ext" else "ref", None)
dev = torch.device("cuda", 0)
bank = synth.word_bank(n_words)
shapes = {"ref": [(0, 0, 0), (5, 100, 16384), (5, 100, 8192), (5, 100, 4096), (4, 100, 8192), (4, 100, 4096), (3, 100, 8192), (3, 100, 4096),
                  (6, 100, 4096), (6, 100, 8192), (7, 100, 4096), (8, 100, 4096), (10, 100, 4096), (4, 50, 8192), (8, 50, 8192), (10, 50, 4096)],
          "ext": [(0, 0, 0), (7, 125, 4096), (4, 125, 8192), (4, 125, 4096), (5, 100, 8192), (5, 100, 16384), (4, 100, 8192), (3, 167, 8192),
                  (6, 84, 8192), (8, 63, 8192), (5, 125, 4096), (6, 125, 4096)]}[which]
pcm = None
for U, kc, g in shapes:
    dev_hook("dtw_u", U)
    dev_hook("dtw_kc", kc)
    dev_hook("dtw_tie_g", g)
    dev_hook("dtw_debug", 1)
    eng = Engine(max_frames=bench.MAX_FRAMES, device=0, **cfg)
    tm, tfr, rng = bench.make_templates(eng, bank, Kt, n_words, rate, dev)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    if pcm is None:
        pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, n_words, B)), [bench.T] * B, seed=1000, bank=bank,
                                    S=synth.buf_len_for(bench.T, rate), device=dev, rate=rate)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    eng.set_pipeline(streams=1)
    eng.recognize_dev(pcm, out)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    for _ in range(3):
        eng.recognize_dev(pcm, out)
    torch.cuda.synchronize()
    st = eng.stage_ms()
    eng.set_profiling(False)
    chk = int(out["scores"].to(torch.int64).sum().item() & 0xFFFFFFFF)
    print(json.dumps({"U": U, "Kc": kc, "tie_g": g, "dtw_ms": round(st["dtw"], 3), "mfcc_ms": round(st["mfcc"], 3), "chk": chk}), flush=True)
    eng.close()
    del out
for h in ("dtw_u", "dtw_kc", "dtw_tie_g", "dtw_debug"):
    dev_hook(h, 0)
