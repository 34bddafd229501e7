"""WAV ingestion: PCM WAV -> the ADC-like unsigned 12-bit codes the engine consumes.

The firmware captures 12-bit right-aligned ADC codes around mid-scale (Src/BSP/ADC.C, ADC.H:7-11); the
reference's own recordings (Matlab/语音样本/*.wav) are 8 kHz, 8- or 16-bit PCM.  Conversion keeps the top 12
bits and re-biases to 2048.  Host-side plumbing only.  Recordings at another rate are brought to the front end's rate
(8 kHz reference, 16 kHz extension) by a polyphase FIR resampler before the 12-bit conversion."""
import wave
from math import gcd

import numpy as np


def pcm_to_adc(samples, sample_width):
    """signed 16-bit (width 2) or unsigned 8-bit (width 1) PCM -> uint16 codes in 0..4095."""
    if sample_width == 2:
        x = (np.asarray(samples, dtype=np.int16).astype(np.int32) >> 4) + 2048
    elif sample_width == 1:
        x = (np.asarray(samples, dtype=np.uint8).astype(np.int32) - 128) * 16 + 2048
    else:
        raise ValueError("only 8- and 16-bit PCM is supported")
    return np.clip(x, 0, 4095).astype(np.uint16)


def resample_pcm(x, rate_in, rate_out, taps_per_phase=24):
    """Polyphase FIR rate conversion of signed samples (float64 in, float64 out), L / M = rate_out / rate_in reduced:
    conceptually zero-stuff by L, low-pass at min(rate_in, rate_out) / 2 with a Kaiser-windowed sinc (beta 8.6), keep
    every M-th sample.  Evaluated in polyphase form: output j sits at position k = half + j*M of the up-sampled stream,
    only the taps h[k mod L + q*L] meet non-zero samples (x[k div L - q]), so an output costs about 2*taps_per_phase*
    max(L, M)/L multiply-adds and nothing of size len(x)*L is ever built (44.1 kHz -> 8 kHz: 265 per output sample).
    numpy only; the filter is linear-phase and its delay is removed."""
    x = np.asarray(x, dtype=np.float64)
    if rate_in == rate_out:
        return x
    g = gcd(int(rate_in), int(rate_out))
    L, M = int(rate_out) // g, int(rate_in) // g
    half = taps_per_phase * max(L, M)
    n = np.arange(-half, half + 1)
    fc = 0.5 / max(L, M)                                        # cycles per sample at the rate rate_in * L
    h = 2 * fc * np.sinc(2 * fc * n) * np.kaiser(len(n), 8.6) * L
    Q = (2 * half) // L + 1                                     # taps that can meet a sample, per output
    hp = np.concatenate([h, np.zeros(L)])                       # tap indices p + q*L may run up to 2*half + L - 1
    xp = np.concatenate([np.zeros(Q), x, np.zeros(Q)])          # sample indices m0 - q run from -Q to len(x) + Q
    n_out = (len(x) * L + M - 1) // M                           # = len(up[::M])
    q = np.arange(Q)
    y = np.empty(n_out, dtype=np.float64)
    for j0 in range(0, n_out, 16384):
        k = half + np.arange(j0, min(n_out, j0 + 16384)) * M
        p, m0 = k % L, k // L
        m = m0[:, None] - q[None, :]
        m = np.where((m >= -Q) & (m < len(x) + Q), m, -Q) + Q   # outside the record: a zero of the padding
        y[j0:j0 + len(k)] = np.einsum("jq,jq->j", hp[p[:, None] + q[None, :] * L], xp[m])
    return y


def wav_to_adc(path, expect_rate=None, resample=False):
    """Read a mono/stereo PCM WAV (first channel) and return uint16 ADC-like codes.  A rate other than expect_rate is
    an error unless resample=True, in which case the signal is converted with resample_pcm()."""
    with wave.open(path, "rb") as w:
        sw, ch, rate, n = w.getsampwidth(), w.getnchannels(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    data = np.frombuffer(raw, dtype=np.int16 if sw == 2 else np.uint8)
    if ch > 1:
        data = data[::ch]
    if expect_rate is not None and rate != expect_rate:
        if not resample:
            raise ValueError(f"{path}: sample rate {rate} Hz, engine front end expects {expect_rate} Hz")
        x = data.astype(np.float64) if sw == 2 else (data.astype(np.float64) - 128.0) * 256.0
        y = np.clip(np.round(resample_pcm(x, rate, expect_rate)), -32768, 32767).astype(np.int16)
        return pcm_to_adc(y, 2)
    return pcm_to_adc(data, sw)


def make_capture(adc, noise_head, buf_len=16000, mid=2048):
    """Capture buffer = noise_head (what the firmware records before the speaker starts, main.c:79-87)
    followed by the audio, padded with mid-scale codes to buf_len."""
    buf = np.full(buf_len, mid, dtype=np.uint16)
    nh = len(noise_head)
    buf[:nh] = noise_head
    m = min(len(adc), buf_len - nh)
    buf[nh:nh + m] = adc[:m]
    return buf
