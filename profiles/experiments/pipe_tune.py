import os, sys, time, numpy as np, torch, subprocess, json
sys.path.insert(0, os.getcwd())
def child():
    from stm32_speech_recognition_amd import Engine, synth
    T, K, NW, B = 256, 100, 25, 65536
    dev = torch.device("cuda", 0)
    eng = Engine(max_frames=320, device=0)
    bank = synth.word_bank(NW); rng = np.random.default_rng(2026)
    tfr = rng.integers(192, 321, K)
    tp = synth.make_utterances(np.arange(K) % NW, tfr, seed=77, bank=bank, S=synth.buf_len_for(320), device=dev)
    tvad, tmf = eng.features_dev(tp); torch.cuda.synchronize()
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    pcm = synth.make_utterances(rng.integers(0, NW, B), [T] * B, seed=1000, bank=bank, S=synth.buf_len_for(T), device=dev)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    res = {}
    for prof in (0, 1):
        eng.recognize_dev(pcm, out); torch.cuda.synchronize()
        eng.set_profiling(bool(prof))
        t0 = time.perf_counter()
        for _ in range(8): eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        res[f"prof{prof}"] = round((time.perf_counter() - t0) / 8 * 1e3, 2)
        eng.set_profiling(False)
    print(json.dumps(res))
if len(sys.argv) > 1:
    child()
else:
    for st, mc, mx, mode in ((1, 4096, 8, 0), (3, 4096, 12, 0), (3, 4096, 9, 0), (3, 4096, 18, 0), (2, 4096, 12, 0), (3, 4096, 6, 0), (3, 2048, 30, 0), (4, 4096, 16, 0), (3, 4096, 12, 0)):
        env = dict(os.environ, SR_PIPE_STREAMS=str(st), SR_PIPE_MIN_CHUNK=str(mc), SR_PIPE_MAX_CHUNKS=str(mx), SR_PIPE_MODE=str(mode))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"mode {mode} streams {st} min_chunk {mc} max_chunks {mx}:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
