"""Generate tests/golden/real_speech.npz: REAL speech through TIER (i) of the oracle (the reference's own VAD.C /
MFCC.C / DTW.C objects, oracle/_ref/libsr_ref.so).

Run in the build container (needs /root/reference):   python tests/golden/make_real_golden.py

Input audio: the reference's own 8 kHz recordings (Matlab/语音样本/*.wav), converted to the 12-bit ADC-like codes the
firmware captures (stm32_speech_recognition_amd/wavio.py) and cut into 16 000-sample capture buffers with a
synthetic 2 400-sample room-noise head (the firmware records the room first, main.c:79-87).  The fixture holds those
capture buffers (data derived from the recordings, not reference source code) and what the reference's compiled code
makes of them; it is what lets the GPU box, where /root/reference does not exist, check parity on real speech:
unlike the synthetic sets, real speech sits right at the u32 wrap of the filterbank products.
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402
from stm32_speech_recognition_amd import wavio  # noqa: E402

SAMPLES = "/root/reference/Matlab/语音样本"
N_CAP, N_TPL = 12, 8


def captures():
    rng = np.random.default_rng(7)
    out = []
    for path in sorted(glob.glob(os.path.join(SAMPLES, "*.wav"))):
        adc = wavio.wav_to_adc(path)
        for off in range(0, max(1, len(adc) - 13600), 27200):
            buf = np.empty(16000, dtype=np.uint16)
            buf[:2400] = np.clip(np.round(2048 + rng.normal(0, 6, 2400)), 0, 4095)
            chunk = adc[off:off + 13600]
            buf[2400:2400 + len(chunk)] = chunk
            buf[2400 + len(chunk):] = 2048
            out.append(buf)
    return out


def main():
    r = ol.RefLib()
    caps = captures()
    # keep captures in which the reference's VAD closes at least one segment that fits the 119-frame record
    good = []
    for buf in caps:
        a, sg = r.vad(buf)
        if sg[1] > 0 and sg[0] >= 1:
            n, m, f = r.mfcc(buf, int(sg[0]), int(sg[1]), a)
            if n:
                good.append(buf)
    assert len(good) >= N_CAP + N_TPL, len(good)
    tpl_caps, test_caps = good[:N_TPL], good[N_TPL:N_TPL + N_CAP]
    # template store = flash image (Flash.H:11-20): save_mdl of segment 0 of the first N_TPL captures
    store = np.full(N_TPL * 4096, 0xFF, dtype=np.uint8)
    for k, buf in enumerate(tpl_caps):
        a, sg = r.vad(buf)
        n, m, f = r.mfcc(buf, int(sg[0]), int(sg[1]), a)
        store[k * 4096:k * 4096 + len(f)] = f
        store[k * 4096:k * 4096 + 2].view(np.uint16)[0] = 12345
    pcm = np.stack(test_caps + tpl_caps[:2])  # two template captures are recognised too (distance 0 to themselves)
    M = len(pcm)
    seg = np.zeros((M, 6), np.int32)
    atap = np.zeros((M, 4), np.uint32)
    st = np.zeros((M, 3), np.uint32)
    best = np.zeros((M, 3), np.uint32)
    dis = np.zeros((M, 3), np.uint32)
    sc = np.zeros((M, 3, N_TPL), np.uint32)
    fr = np.zeros((M, 3), np.uint32)
    mf = np.zeros((M, 3, 119, 12), np.int16)
    for i in range(M):
        a, sg = r.vad(pcm[i])
        seg[i], atap[i] = sg, a.astuple()
        for s_ in range(3):
            st[i, s_], best[i, s_], dis[i, s_], sc[i, s_], m, fr[i, s_] = r.spch_recg(pcm[i], store, seg_idx=s_)
            if st[i, s_] != 0:
                sc[i, s_] = 0xFFFFFFFF
                fr[i, s_] = 0
            else:
                mf[i, s_, :fr[i, s_]] = m
    out = os.path.join(HERE, "real_speech.npz")
    np.savez_compressed(out, pcm=pcm, store=store, seg=seg, atap=atap, status=st, best=best, dis=dis, scores=sc, frm=fr,
                        mfcc=mf)
    print("wrote", out, os.path.getsize(out), "bytes; status", st.tolist(), "frm", fr.tolist())


if __name__ == "__main__":
    main()
