"""Exhaustive check (all 2^32 inputs) of the path's non-integer device functions against the host C expressions:
(u32)(log((double)x)*100) [MFCC.C:168], (u32)sqrtf((float)x) [DTW.C:59], (u32)(sqrtf((float)(s32)x)*10) [MFCC.C:56-58].
Not part of the pytest run (about a minute on a many-core GPU box):
    python tests/exhaustive_math_sweep.py  ->  profiles/exhaustive_math_sweep.txt
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from stm32_speech_recognition_amd import Engine  # noqa: E402
from stm32_speech_recognition_amd.engine import _vp  # noqa: E402


def main():
    eng = Engine(device=0)
    orc = ol.Oracle()
    threads = min(512, os.cpu_count() or 8)
    chunk = 1 << 25
    got = np.zeros(3 * chunk, np.uint32)
    want = np.zeros(3 * chunk, np.uint32)
    bad = 0
    t0 = time.time()
    for c in range((1 << 32) // chunk):
        x = (np.arange(chunk, dtype=np.uint64) + np.uint64(c) * np.uint64(chunk)).astype(np.uint32)
        assert eng.L.sr_math_diag(eng.h, _vp(x), _vp(got), C.c_uint32(chunk)) == 0
        orc.L.sr_oracle_math_diag_mt(x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_uint32(chunk),
                                     C.c_uint32(threads))
        nb = int((got != want).sum())
        if nb:
            i = np.nonzero(got != want)[0][:5]
            print("MISMATCH chunk", c, x[i // 3], i % 3, got[i], want[i], flush=True)
        bad += nb
    msg = (f"exhaustive sweep of 2^32 inputs x 3 functions (log*100, sqrtf, sqrtf*10): {bad} mismatches, "
           f"{time.time() - t0:.0f} s, {threads} host threads, device gfx950")
    print(msg)
    out = os.path.join(ROOT, "gpurun_out", "exhaustive_math_sweep.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write(msg + "\n")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
