"""Synthetic 8 kHz / 12-bit-ADC capture buffers for tests and bench.

A capture buffer mirrors what the firmware's record() leaves in VcBuf (reference
Src/APP/main.c:77-102, Src/BSP/ADC.H:7-11): unsigned 12-bit ADC codes around a
mid-scale DC level, a noise-only head (atap_len = 2400 samples) that noise_atap
adapts to, then quiet, one spoken "word", and quiet again.

Layout for T target frames (frame 160, hop 80):

    [0, 2400)            noise head          N(0, head_sigma)
    [2400, 3280)         quiet               N(0, quiet_sigma)
    [3280, 3280+80(T-1)) word                3 chirped sinusoids * envelope + N(0, 30)
    [.., +1680)          quiet

The endpoint detector (VAD.C:164-215) then yields the segment
[3200, 3200 + 80(T-1) + 160), i.e. exactly T frames, as long as the quiet parts stay
below the adaptive thresholds (quiet_sigma < head_sigma gives a wide margin; with
quiet_sigma == head_sigma the tight s_thl_ratio 11/10 lets noise frames re-trigger
and T varies -- useful for parity tests, not for the fixed-T benchmark).

A "word" is a set of 3 frequency tracks (piecewise-linear in normalised time between
5 control points).  Utterances of the same word share the tracks and differ in phase,
amplitude, duration and noise, so greedy-DTW matching against word templates is a
meaningful classification task.

torch is used as the array library so the same code fills host buffers (tests) and
HBM-resident buffers (bench) -- data plumbing, not part of the recognition path.
"""
import math

import torch

FS = 8000
FRAME = 160
HOP = 80
NOISE_LEN = 2400
LEAD_QUIET = 800
TAIL_QUIET = 1600
MID = 2048
N_CTRL = 5


def buf_len_for(T, rate=1):
    """Capture-buffer length that holds a T-frame word (25 360 for T = 256).  rate = 2: the 16 kHz extension
    front end (every length doubles: 320/160 framing, 4800-sample noise head)."""
    return rate * (NOISE_LEN + LEAD_QUIET + HOP * (T - 1) + FRAME + TAIL_QUIET)


def word_bank(n_words, seed=1234):
    """Frequency tracks [n_words, 3, N_CTRL] in Hz (200..3000) and base amplitudes [n_words, 3]."""
    g = torch.Generator().manual_seed(seed)
    f = 200.0 + 2800.0 * torch.rand(n_words, 3, N_CTRL, generator=g)
    a = 80.0 + 170.0 * torch.rand(n_words, 3, generator=g)
    return f, a


def make_utterances(word_ids, frames, seed, bank, S=None, gain=1.0, head_sigma=8.0, quiet_sigma=4.0,
                    speech_sigma=30.0, device="cpu", chunk=2048, rate=1):
    """uint16-valued capture buffers, returned as an int16 torch tensor view-compatible with u16 [B, S].

    word_ids: int64 [B]; frames: int64 [B] target frame counts (<= the T that S was sized for).
    Returns torch.int16 tensor [B, S] holding the raw 16-bit ADC codes (values 0..4095, so the
    int16 bit pattern equals the uint16 one; torch has no uint16 arithmetic).
    """
    word_ids = torch.as_tensor(word_ids, dtype=torch.int64)
    frames = torch.as_tensor(frames, dtype=torch.int64)
    B = word_ids.numel()
    if S is None:
        S = buf_len_for(int(frames.max()), rate)
    hop, noise_len, fs = HOP * rate, NOISE_LEN * rate, FS * rate
    f_bank, a_bank = bank
    out = torch.empty(B, S, dtype=torch.int16, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(S, device=device, dtype=torch.float32)
    p0 = rate * (NOISE_LEN + LEAD_QUIET + HOP)
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        n = b1 - b0
        wid = word_ids[b0:b1]
        fr = frames[b0:b1].to(device=device, dtype=torch.float32)
        span = (hop * (fr - 1)).clamp(min=1).unsqueeze(1)         # speech samples (a 1-frame request has no span)
        u = ((t.unsqueeze(0) - p0) / span).clamp(0.0, 1.0)        # normalised time [n, S]
        in_word = (t.unsqueeze(0) >= p0) & (t.unsqueeze(0) < p0 + span)
        fc = f_bank[wid].to(device)                               # [n, 3, N_CTRL]
        amp = a_bank[wid].to(device) * (0.8 + 0.4 * torch.rand(n, 3, generator=g, device=device)) * gain
        ph0 = 2 * math.pi * torch.rand(n, 3, generator=g, device=device)
        sig = torch.zeros(n, S, device=device)
        seg = u * (N_CTRL - 1)
        i0 = seg.floor().clamp(max=N_CTRL - 2).to(torch.int64)
        w = seg - i0
        for k in range(3):
            f0 = torch.gather(fc[:, k, :], 1, i0)
            f1 = torch.gather(fc[:, k, :], 1, i0 + 1)
            f = f0 + (f1 - f0) * w
            # phase = integral of f; float64 accumulate keeps the chirp clean over 20k samples
            ph = torch.cumsum((f * in_word).to(torch.float64), dim=1) * (2 * math.pi / fs)
            sig += amp[:, k:k + 1] * torch.sin(ph.to(torch.float32) + ph0[:, k:k + 1])
        env = 0.55 + 0.45 * torch.sin(math.pi * u) ** 2           # floor keeps the word's edges loud
        noise = torch.randn(n, S, generator=g, device=device)
        sigma = torch.where(in_word, torch.tensor(speech_sigma, device=device),
                            torch.where(t.unsqueeze(0) < noise_len, torch.tensor(head_sigma, device=device),
                                        torch.tensor(quiet_sigma, device=device)))
        x = MID + torch.where(in_word, sig * env, torch.zeros_like(sig)) + noise * sigma
        out[b0:b1] = x.round().clamp(0, 4095).to(torch.int16)
    return out


def make_multiword(word_ids, frames, seed, bank, S=16000, gap=1200, **kw):
    """One capture buffer holding several words separated by `gap` quiet samples (>= 11 quiet frames so the
    VAD closes each segment, VAD.C:198-206).  Returns an int16 tensor [S] of ADC codes."""
    g = torch.Generator().manual_seed(seed)
    out = (MID + 4.0 * torch.randn(S, generator=g)).round().clamp(0, 4095).to(torch.int16)
    out[:NOISE_LEN] = (MID + 8.0 * torch.randn(NOISE_LEN, generator=g)).round().clamp(0, 4095).to(torch.int16)
    pos = NOISE_LEN + LEAD_QUIET + HOP
    for i, (w, t) in enumerate(zip(word_ids, frames)):
        one = make_utterances([w], [t], seed=seed * 31 + i, bank=bank, S=buf_len_for(t), **kw)[0]
        p0 = NOISE_LEN + LEAD_QUIET + HOP
        span = HOP * (t - 1)
        if pos + span + gap > S:
            break
        out[pos:pos + span] = one[p0:p0 + span]
        pos += span + gap
    return out


def as_u16_numpy(x):
    """int16 torch tensor of ADC codes -> numpy uint16 (same bits)."""
    import numpy as np
    return x.cpu().numpy().view(np.uint16)
