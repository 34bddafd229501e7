import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle_lib as ol
from stm32_speech_recognition_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|MHz' | head -8")
NW, K, T = 25, 100, 256
bank = synth.word_bank(NW); rng = np.random.default_rng(1)
orc = ol.Oracle(max_frames=320)
tfr = rng.integers(192, 321, K)
tp = synth.as_u16_numpy(synth.make_utterances(np.arange(K) % NW, tfr, seed=77, bank=bank, S=synth.buf_len_for(320)))
dummy = orc.make_templates(np.zeros((1, 321, 12), np.int16), np.array([10], np.uint32))
_, mf, _ = orc.recognize_batch(tp, dummy, n_threads=32, want_scores=False)
tm = np.concatenate([mf[:, :320], np.zeros((K, 1, 12), np.int16)], 1)
tpl = orc.make_templates(tm, tfr.astype(np.uint32))
n = 2048
pcm = synth.as_u16_numpy(synth.make_utterances(rng.integers(0, NW, n), [T] * n, seed=3, bank=bank, S=synth.buf_len_for(T)))
for th in (1, 4, 16, 64, 128, 256):
    m = min(n, max(16, th * 8))
    t0 = time.perf_counter(); orc.recognize_batch(pcm[:m], tpl, n_threads=th, want_mfcc=False, want_scores=True); dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {m} utt in {dt:.3f} s = {m/dt:.0f} utt/s = {m/dt/th:.1f} per thread", flush=True)
