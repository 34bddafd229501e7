"""Throughput of the GENERIC front end next to the reference front end on the same captures (development aid).
    python profiles/experiments/generic_rate.py [B]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd.engine import vad_from_torch

def run(B, K, **cfg):
    dev = torch.device("cuda", 0)
    eng = Engine(max_frames=320, device=0, **cfg)
    bank = synth.word_bank(20)
    rng = np.random.default_rng(1)
    tfr = rng.integers(192, 321, K)
    tp = synth.make_utterances(np.arange(K) % 20, tfr, seed=77, bank=bank, S=synth.buf_len_for(320), device=dev)
    tv, tm = eng.features_dev(tp)
    torch.cuda.synchronize()
    v = vad_from_torch(tv)
    assert (v["status"] == 0).all()
    nc = eng.n_coef
    tmh = np.concatenate([tm.cpu().numpy(), np.zeros((K, 1, nc), np.int16)], 1)
    eng.set_templates_dense(tmh, v["frm_num"].astype(np.uint32))
    pcm = synth.make_utterances(torch.from_numpy(rng.integers(0, 20, B)), [256] * B, seed=5, bank=bank, S=synth.buf_len_for(256), device=dev)
    out = eng.alloc_outputs(B, dev)
    eng.recognize_dev(pcm, out)
    torch.cuda.synchronize()
    eng.set_pipeline(streams=1)
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.recognize_dev(pcm, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = eng.stage_ms()
    eng.close()
    return {"utt_per_s": B / dt, "ms": dt * 1e3, "kernel_ms": {k: round(st[k], 3) for k in ("vad", "mfcc", "dtw", "argmin")}}

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
res = {"B": B, "K": 100,
       "reference (8 kHz, 24 Mel, 12 coef: k_mfcc + k_dtw_lds)": run(B, 100),
       "generic 8 kHz, 26 Mel, 12 coef (k_mfcc_gen + k_dtw_lds)": run(B, 100, n_mel=26),
       "generic 8 kHz, 26 Mel, 10 coef (k_mfcc_gen + k_dtw_lds on rows zero-padded to 12)": run(B, 100, n_mel=26, n_coef=10),
       "generic 8 kHz, 26 Mel, 13 coef (k_mfcc_gen + k_dtw_lds, 16-wide rows)": run(B, 100, n_mel=26, n_coef=13)}
print(json.dumps(res, indent=1))
