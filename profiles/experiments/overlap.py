import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from stm32_speech_recognition_amd import Engine, synth
from stm32_speech_recognition_amd.engine import vad_from_torch, results_from_torch
T, K, NW, B = 256, 100, 25, 65536
dev = torch.device("cuda", 0)
eng = Engine(max_frames=320, device=0)
bank = synth.word_bank(NW); rng = np.random.default_rng(2026)
tfr = rng.integers(192, 321, K)
tp = synth.make_utterances(np.arange(K) % NW, tfr, seed=77, bank=bank, S=synth.buf_len_for(320), device=dev)
tvad, tmf = eng.features_dev(tp); torch.cuda.synchronize()
tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
eng.set_templates_dense(tm, tfr.astype(np.uint32))
S = synth.buf_len_for(T)
pcm = synth.make_utterances(rng.integers(0, NW, B), [T] * B, seed=1000, bank=bank, S=S, device=dev)
def run(nsplit, nstreams, steps=6):
    outs = [eng.alloc_outputs(B // nsplit, dev, mfcc=True, vad=True) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    parts = [pcm[i * (B // nsplit):(i + 1) * (B // nsplit)] for i in range(nsplit)]
    def step():
        for i in range(nsplit):
            with torch.cuda.stream(streams[i % nstreams]):
                eng.recognize_dev(parts[i], outs[i])
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r = torch.cat([o["results"] for o in outs]).cpu().numpy()
    return dt * 1e3, r
base, r0 = run(1, 1)
print(f"1 split 1 stream: {base:.2f} ms")
for ns, nst in ((2, 1), (2, 2), (4, 2), (4, 4), (8, 2), (8, 4), (16, 4)):
    ms, r = run(ns, nst)
    print(f"{ns} splits {nst} streams: {ms:.2f} ms  same={np.array_equal(r, r0)}", flush=True)
