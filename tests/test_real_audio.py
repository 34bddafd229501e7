"""Real speech through both oracle tiers (CPU, only where /root/reference exists).

The reference ships 13 recordings (Matlab/语音样本/*.wav, 8 kHz, 8/16-bit) and three raw 12-bit ADC dumps.
They are read in place, converted to the ADC-like 12-bit codes the
firmware captures (ADC.C: 12-bit right-aligned, mid-scale ~2048) and pushed through
noise_atap -> VAD -> get_mfcc -> dtw with the reference's own objects (tier i) and with the parametrised
restatement (tier ii).  Everything must be bit-identical, which pins the restatement on real signals, not
only on synthetic ones."""
import glob
import os
import wave

import numpy as np

import oracle_lib as ol
from conftest import REFERENCE, needs_reference
from stm32_speech_recognition_amd import wavio

SAMPLES = os.path.join(REFERENCE, "Matlab", "语音样本")


def _captures():
    """16 000-sample capture buffers cut from the reference's recordings: 2 400 samples of low-level noise
    (the firmware records the room before the speaker starts, main.c:79-87) + 13 600 samples of audio."""
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
            if len(out) >= 40:
                return out
    return out


@needs_reference
def test_wav_reader_handles_the_reference_recordings():
    paths = sorted(glob.glob(os.path.join(SAMPLES, "*.wav")))
    assert len(paths) >= 10
    for p in paths:
        with wave.open(p, "rb") as w:
            assert w.getframerate() == 8000
        adc = wavio.wav_to_adc(p)
        assert adc.dtype == np.uint16 and adc.max() <= 4095 and len(adc) > 8000


@needs_reference
def test_real_speech_tier2_equals_tier1():
    r = ol.RefLib()
    o = ol.Oracle(max_frames=119)
    caps = _captures()
    assert len(caps) >= 20
    feats, nseg = [], 0
    for buf in caps:
        a1, s1 = r.vad(buf)
        rc, a2 = o.noise_atap(buf)
        s2 = o.vad(buf, a2)
        assert a1.astuple() == a2.astuple() and np.array_equal(s1, s2)
        for k in range(3):
            st, en = int(s1[2 * k]), int(s1[2 * k + 1])
            if en < 0 or st < 1:
                continue
            nseg += 1
            n1, m1, f1 = r.mfcc(buf, st, en, a1)
            n2, m2 = o.mfcc(buf, st, en, a2)
            assert n1 == n2 and np.array_equal(m1, m2)
            if n1:
                feats.append((n1, m1, f1))
    assert nseg >= 15 and len(feats) >= 10
    pad = np.zeros((2, 12), dtype=np.int16)
    feats = feats[:24]
    for i in range(len(feats)):
        for j in range(len(feats)):
            d1 = r.dtw(feats[i][2], feats[j][2])
            d2 = o.dtw(np.concatenate([feats[i][1], pad]), feats[i][0], np.concatenate([feats[j][1], pad]), feats[j][0])
            assert d1 == d2


def test_resampler_keeps_a_tone_and_its_level(tmp_path):
    """wavio.resample_pcm / wav_to_adc(resample=True): a 440 Hz tone recorded at 44.1, 22.05, 16 and 11.025 kHz arrives at
    8 kHz with its frequency, its amplitude (within 1 %) and its length; an 8 kHz file passes through untouched."""
    for rate in (44100, 22050, 16000, 11025, 8000):
        t = np.arange(int(rate * 0.5)) / rate
        x = np.round(12000 * np.sin(2 * np.pi * 440 * t)).astype(np.int16)
        path = str(tmp_path / f"tone_{rate}.wav")
        with wave.open(path, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(rate)
            w.writeframes(x.tobytes())
        adc = wavio.wav_to_adc(path, expect_rate=8000, resample=True).astype(np.float64) - 2048
        assert abs(len(adc) - 4000) <= 1
        mid = adc[400:3600]
        spec = np.abs(np.fft.rfft(mid * np.hanning(len(mid))))
        assert abs(np.argmax(spec) * 8000 / len(mid) - 440) < 4
        assert abs(np.sqrt(2 * np.mean(mid ** 2)) - 12000 / 16) < 0.01 * 12000 / 16
    import pytest
    with pytest.raises(ValueError):
        wavio.wav_to_adc(str(tmp_path / "tone_16000.wav"), expect_rate=8000)
