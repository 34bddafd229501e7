"""rocprofv3 --kernel-trace --stats -- python profiles/experiments/small_launch_trace.py: true durations of the small-launch kernel
forms (k_vad_wide, k_mfcc<2>, k_dtw_cells) in a loop of 200 single-capture calls at the firmware shapes (16 000-sample capture,
110-frame word, 80 slots); profiles/r04_small_launch_rocprof.csv is its summary."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from stm32_speech_recognition_amd import Engine, synth
Tl, S, Kl = 110, 16000, 80
eng = Engine(max_frames=119, device=0)
bank = synth.word_bank(25)
rng = np.random.default_rng(4)
tfr = rng.integers(70, 120, Kl)
tp = synth.as_u16_numpy(synth.make_utterances(np.arange(Kl) % 25, tfr, seed=8, bank=bank, S=S))
store, st = eng.train_store(tp, np.arange(Kl), n_slots=Kl)
eng.set_templates_store(store)
dpcm = synth.make_utterances(rng.integers(0, 25, 4), [Tl] * 4, seed=9, bank=bank, S=S, device=torch.device("cuda", 0))
o = eng.alloc_outputs(1, "cuda:0", mfcc=False, vad=False)
for i in range(200):
    eng.recognize_dev(dpcm[:1], o); torch.cuda.synchronize()
