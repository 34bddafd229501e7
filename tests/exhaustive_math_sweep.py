"""Exhaustive check (all 2^32 inputs) of the path's non-integer device functions against the host C expressions:
(u32)(log((double)x)*100) [MFCC.C:168], (u32)sqrtf((float)x) [DTW.C:59], (u32)(sqrtf((float)(s32)x)*10) [MFCC.C:56-58].
Not part of the pytest run (about a minute on a many-core GPU box):
    python tests/exhaustive_math_sweep.py  ->  profiles/exhaustive_math_sweep.txt
plus (round 5; --mel-only runs just this, seconds) the fused Mel filterbank term of k_mfcc over its whole certified domain:
    every weight 0..1000 x every energy 0..floor(2^28/100)  ->  profiles/rNN_mel_term_sweep.txt
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
    for c in range(0 if "--mel-only" in sys.argv else (1 << 32) // chunk):
        x = (np.arange(chunk, dtype=np.uint64) + np.uint64(c) * np.uint64(chunk)).astype(np.uint32)
        assert eng.L.sr_math_diag(eng.h, _vp(x), _vp(got), C.c_uint32(chunk)) == 0
        orc.L.sr_oracle_math_diag_mt(x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_uint32(chunk),
                                     C.c_uint32(threads))
        nb = int((got != want).sum())
        if nb:
            i = np.nonzero(got != want)[0][:5]
            print("MISMATCH chunk", c, x[i // 3], i % 3, got[i], want[i], flush=True)
        bad += nb
    # the fused Mel filterbank term of k_mfcc (round 5): every weight 0..1000 x every energy 0..floor(2^28/100)
    emax = (1 << 28) // 100
    mel_bad = np.zeros(1001, np.uint64)
    t1 = time.time()
    assert eng.L.sr_mel_term_sweep(eng.h, C.c_uint32(0), C.c_uint32(1001), C.c_uint32(emax), _vp(mel_bad)) == 0
    mel_msg = (f"fused Mel filterbank term mul_hi(E << 4, ceil(tri * 2^28 / 100)) vs the reference's u32 E*tri/100 (MFCC.C:139-161): "
               f"1001 weights (0..1000) x {emax + 1} energies (0..{emax}) = {1001 * (emax + 1)} terms, "
               f"{int(mel_bad.sum())} mismatches, weight recovery mul_hi(M, 1600) == tri for all, {time.time() - t1:.1f} s on the device")
    print(mel_msg)
    bad += int(mel_bad.sum())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "mel_term_sweep.txt"), "w").write(mel_msg + "\n")
    if "--mel-only" in sys.argv:
        return 0 if bad == 0 else 1
    msg = (f"exhaustive sweep of 2^32 inputs x 3 functions (log*100, sqrtf, sqrtf*10): {bad} mismatches, "
           f"{time.time() - t0:.0f} s, {threads} host threads, device gfx950")
    print(msg)
    out = os.path.join(ROOT, "gpurun_out", "exhaustive_math_sweep.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write(msg + "\n")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
