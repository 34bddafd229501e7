"""How often k_dtw_lds leaves its fast path (library built with -DSR_DTW_STATS: profiles/experiments/ab_build.sh stats . -DSR_DTW_STATS).
    SR_ENGINE_LIB=ab_libs/stats.so python profiles/experiments/dtw_stats.py [ref|ext] [gain]
Counters (wave-steps): total, any lane unsafe, root one short (m2 >= T), root beyond the staged table, a lost lane; lane-steps."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd import engine as E

which = sys.argv[1] if len(sys.argv) > 1 else "ref"
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
B = 8192
rate, cfg, Kt, n_words = bench.workload_setup(which, None)
dev = torch.device("cuda", 0)
bank = synth.word_bank(n_words)
eng = Engine(max_frames=bench.MAX_FRAMES, device=0, **cfg)
tm, tfr, rng = bench.make_templates(eng, bank, Kt, n_words, rate, dev, gain)
eng.set_templates_dense(tm, tfr.astype(np.uint32))
pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, n_words, B)), [bench.T] * B, seed=1000, bank=bank,
                            S=synth.buf_len_for(bench.T, rate), device=dev, rate=rate, gain=gain)
out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
eng.set_pipeline(streams=1)
L = E.load_library()
st = (ctypes.c_ulonglong * 8)()
L.sr_debug_dtw_stats(st, 1)
eng.recognize_dev(pcm, out)
torch.cuda.synchronize()
L.sr_debug_dtw_stats(st, 1)
v = list(st)
sc = out["scores"].cpu().numpy().astype(np.int64)
print(json.dumps({"workload": which, "gain": gain, "wave_steps": v[0], "unsafe": v[1], "short": v[2], "beyond_table": v[3], "lost": v[4],
                  "lane_steps": v[5], "frac_unsafe": round(v[1] / v[0], 5), "frac_short": round(v[2] / v[0], 5),
                  "frac_beyond": round(v[3] / v[0], 5), "frac_lost": round(v[4] / v[0], 5),
                  "score_median": float(np.median(sc[sc < 0xFFFFFFFF])), "score_p99": float(np.percentile(sc[sc < 0xFFFFFFFF], 99))}))
