import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from stm32_speech_recognition_amd import Engine
from stm32_speech_recognition_amd.engine import _vp
eng = Engine(device=0)
chunk = 1 << 25
got = np.zeros(3 * chunk, np.uint32)
hist = np.zeros(3, np.int64); hist_sq = np.zeros(3, np.int64)
for c in range((1 << 32) // chunk):
    x = (np.arange(chunk, dtype=np.uint64) + np.uint64(c) * np.uint64(chunk)).astype(np.uint32)
    assert eng.L.sr_math_diag(eng.h, _vp(x), _vp(got), C.c_uint32(chunk)) == 0
    d = got[2::3]
    assert d.max() <= 2
    hist += np.bincount(d, minlength=3)
    if c == 0:
        k = np.arange(0, 5792, dtype=np.uint32); sq = k * k
        hist_sq += np.bincount(d[sq], minlength=3)
print("v_sqrt_f32 - RN sqrt in ulps [-1,0,+1]:", hist, " perfect squares < 2^25:", hist_sq)
