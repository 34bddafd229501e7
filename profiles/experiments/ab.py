"""A/B two builds of libsr_engine.so on the same box: python profiles/experiments/ab.py libA.so libB.so [rounds]"""
import os, sys, time, json, subprocess
import numpy as np
def child():
    import torch
    sys.path.insert(0, os.getcwd())
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
    for name, streams in (("pipe", 3), ("iso", 1)):
        eng.set_pipeline(streams=streams)
        eng.recognize_dev(pcm, out); torch.cuda.synchronize()
        eng.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(6): eng.recognize_dev(pcm, out)
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / 6 * 1e3, 3)
        st = eng.stage_ms(); eng.set_profiling(False)
        if streams == 1: res.update({k: round(st[k], 3) for k in ("vad", "mfcc", "dtw")})
    res["chk"] = int(out["scores"].to(torch.int64).sum().item() & 0xFFFFFFFF)
    print(json.dumps(res))
if sys.argv[1] == "child":
    child()
else:
    libs = sys.argv[1:3]; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, SR_ENGINE_LIB=os.path.abspath(lib))
            p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(os.path.basename(lib), p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:], flush=True)
