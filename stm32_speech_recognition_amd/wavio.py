"""WAV ingestion: PCM WAV -> the ADC-like unsigned 12-bit codes the engine consumes.

The firmware captures 12-bit right-aligned ADC codes around mid-scale (Src/BSP/ADC.C, ADC.H:7-11); the
reference's own recordings (Matlab/语音样本/*.wav) are 8 kHz, 8- or 16-bit PCM.  Conversion keeps the top 12
bits and re-biases to 2048.  Host-side plumbing only; no resampling (the sample rate must already match the
engine's front end: 8 kHz reference, 16 kHz extension)."""
import wave

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


def wav_to_adc(path, expect_rate=None):
    """Read a mono/stereo PCM WAV (first channel) and return uint16 ADC-like codes."""
    with wave.open(path, "rb") as w:
        sw, ch, rate, n = w.getsampwidth(), w.getnchannels(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if expect_rate is not None and rate != expect_rate:
        raise ValueError(f"{path}: sample rate {rate} Hz, engine front end expects {expect_rate} Hz")
    data = np.frombuffer(raw, dtype=np.int16 if sw == 2 else np.uint8)
    if ch > 1:
        data = data[::ch]
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
