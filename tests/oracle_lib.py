"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package never does.

  Oracle  -- tier (ii): oracle/liboracle.so, own parametrised restatement
  RefLib  -- tier (i):  oracle/_ref/libsr_ref.so, the reference's own VAD.C /
             MFCC.C / DTW.C compiled verbatim (+ C transcription of the asm FFT)
  RefLib320 -- the same objects built with vv_tim_max = 3210 ms (320 frames instead of
             119; oracle/Makefile), so the reference's own code runs the benchmark shape
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_PATH = os.path.join(ORACLE_DIR, "_ref", "libsr_ref.so")
REF320_PATH = os.path.join(ORACLE_DIR, "_ref", "libsr_ref320.so")   # same objects, vv_tim_max patched to 320 frames

DIS_ERR = 0xFFFFFFFF
ST_OK, ST_VAD_FAIL, ST_MFCC_FAIL, ST_SEG_OOB = 0, 1, 2, 3


def build(force=False):
    """(Re)build the oracle libraries with oracle/Makefile (gcc only, seconds)."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("sr_oracle.c", "sr_oracle.h", "q15_fft.c", "ref_glue.c", "Makefile")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s))
    if os.path.isdir("/root/reference/Src/Speech_Recog") and not (os.path.exists(REF_PATH) and os.path.exists(REF320_PATH)):
        stale = True
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)


class Cfg(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("fs", "frame_time", "frame_mov_t", "nfft", "n_mel", "n_coef",
                                          "max_frames", "noise_len_t", "max_seg")]


class Atap(C.Structure):
    _fields_ = [("mid_val", C.c_uint32), ("n_thl", C.c_uint16), ("z_thl", C.c_uint16), ("s_thl", C.c_uint32)]

    def astuple(self):
        return (self.mid_val, self.n_thl, self.z_thl, self.s_thl)


class Templates(C.Structure):
    _fields_ = [("mfcc", C.c_void_p), ("frames", C.c_void_p), ("valid", C.c_void_p),
                ("n", C.c_uint32), ("stride", C.c_uint32)]


RESULT_DTYPE = np.dtype([("best_tpl", "<u4"), ("min_dis", "<u4"), ("frm_num", "<u4"), ("status", "<u4")])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# configurations of the GENERIC front end (round 4: the reference's compile-time constants as run-time values) that the
# tests run; none is one of the two specialised front ends
GENERIC_CONFIGS = [
    # engine keywords                                                              oracle keywords
    (dict(n_mel=26, n_coef=13),                                                    dict(n_mel=26, n_coef=13)),
    (dict(fs=16000, n_mel=40),                                                     dict(fs=16000, n_mel=40)),
    (dict(frame_time_ms=32, frame_mov_ms=16, n_mel=20, n_coef=10, noise_len_ms=480),
     dict(frame_time=32, frame_mov_t=16, n_mel=20, n_coef=10, noise_len_t=480)),
    (dict(frame_time_ms=30, frame_mov_ms=15, n_coef=16),                           dict(frame_time=30, frame_mov_t=15, n_coef=16)),
    (dict(fs=16000, frame_time_ms=32, frame_mov_ms=16, n_mel=64, n_coef=8, noise_len_ms=480),
     dict(fs=16000, frame_time=32, frame_mov_t=16, n_mel=64, n_coef=8, noise_len_t=480)),
    # the edges of the accepted ranges: the fewest filters / one coefficient at an odd rate; the longest frame, most filters
    (dict(fs=12000, n_mel=4, n_coef=1),
     dict(fs=12000, n_mel=4, n_coef=1)),
    (dict(fs=20000, n_mel=30, n_coef=12),
     dict(fs=20000, n_mel=30, n_coef=12)),
    (dict(frame_time_ms=64, frame_mov_ms=32, n_mel=64, n_coef=16, noise_len_ms=960),
     dict(frame_time=64, frame_mov_t=32, n_mel=64, n_coef=16, noise_len_t=960)),
    # hops above 40 ms: the duration limits of VAD.C:72-75 become ONE frame (80 / 64 = 110 / 64 = 1), where the onset / tail
    # counters are still only checked on the second frame (VAD.C:173-181, 196-207)
    (dict(fs=4000, frame_time_ms=128, frame_mov_ms=64, n_mel=12, n_coef=6, noise_len_ms=1920),
     dict(fs=4000, frame_time=128, frame_mov_t=64, n_mel=12, n_coef=6, noise_len_t=1920)),
]


class Oracle:
    def __init__(self, max_frames=119, **kw):
        build()
        L = C.CDLL(LIB_PATH)
        self.L = L
        cfg = Cfg()
        L.sr_oracle_default_cfg(C.byref(cfg))
        cfg.max_frames = max_frames
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        L.sr_oracle_create.restype = C.c_void_p
        self.h = L.sr_oracle_create(C.byref(cfg))
        if not self.h:
            raise ValueError("unsupported oracle config")
        self.h = C.c_void_p(self.h)
        for f in ("frame_len", "hop", "noise_len"):
            getattr(L, "sr_oracle_" + f).restype = C.c_uint32
            getattr(L, "sr_oracle_" + f).argtypes = [C.c_void_p]
        self.frame_len = L.sr_oracle_frame_len(self.h)
        self.hop = L.sr_oracle_hop(self.h)
        self.noise_len = L.sr_oracle_noise_len(self.h)
        self.n_coef = cfg.n_coef
        self.n_mel = cfg.n_mel
        self.max_frames = cfg.max_frames
        L.sr_oracle_get_dis.restype = C.c_uint32
        L.sr_oracle_dtw.restype = C.c_uint32
        L.sr_oracle_mfcc.restype = C.c_uint32

    def __del__(self):
        try:
            self.L.sr_oracle_destroy(self.h)
        except Exception:
            pass

    def _tab(self, name, n, dt):
        f = getattr(self.L, "sr_oracle_" + name)
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p]
        return np.ctypeslib.as_array(C.cast(f(self.h), C.POINTER(dt)), shape=(n,)).copy()

    def tables(self):
        nb = self.cfg.nfft // 2
        return dict(hamm=self._tab("hamm", self.frame_len, C.c_uint16),
                    tri_cen=self._tab("tri_cen", self.n_mel, C.c_uint16),
                    tri_odd=self._tab("tri_odd", nb, C.c_uint16),
                    tri_even=self._tab("tri_even", nb, C.c_uint16),
                    dct=self._tab("dct", self.n_coef * self.n_mel, C.c_int8))

    def noise_atap(self, pcm, n_len=None):
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        a = Atap()
        rc = self.L.sr_oracle_noise_atap(self.h, _p(pcm), C.c_uint32(self.noise_len if n_len is None else n_len),
                                         C.byref(a))
        return rc, a

    def vad(self, pcm, atap):
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        seg = np.full(2 * self.cfg.max_seg, -1, dtype=np.int32)
        self.L.sr_oracle_vad(self.h, _p(pcm), C.c_uint32(len(pcm)), C.byref(atap), _p(seg))
        return seg

    def fft_mag(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.int16)
        mag = np.zeros(self.cfg.nfft // 2, dtype=np.uint32)
        self.L.sr_oracle_fft_mag(self.h, _p(frame), C.c_uint32(len(frame)), _p(mag))
        return mag

    def mfcc(self, pcm, start, end, atap):
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        out = np.zeros((self.max_frames + 1, self.n_coef), dtype=np.int16)
        n = self.L.sr_oracle_mfcc(self.h, _p(pcm), C.c_int32(start), C.c_int32(end), C.byref(atap), _p(out))
        return n, out[:n].copy()

    def frame_peaks(self, pcm, start, end, atap):
        """diagnostic: largest re^2 + im^2 of every frame of the segment (which tier of k_mfcc a frame falls into)"""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        out = np.zeros(self.max_frames + 1, dtype=np.uint32)
        self.L.sr_oracle_frame_peaks.restype = C.c_uint32
        n = self.L.sr_oracle_frame_peaks(self.h, _p(pcm), C.c_int32(start), C.c_int32(end), C.byref(atap), _p(out),
                                         C.c_uint32(len(out)))
        return out[:n].copy()

    def frame_tiers(self, pcm_batch):
        """fractions of the frames of segment 0 of each capture in k_mfcc's QUIET / MID / LOUD tier (csrc/sr_tables.h:
        kMagSmallMax = 26 843, kMagCheapMax = 70 171) and the number of frames counted"""
        pk = []
        for row in pcm_batch:
            rc, a = self.noise_atap(row)
            seg = self.vad(row, a)
            if rc == 0 and seg[0] >= 0 and seg[1] >= 0:
                pk.append(self.frame_peaks(row, int(seg[0]), int(seg[1]), a))
        pk = np.concatenate(pk) if pk else np.zeros(0, np.uint32)
        n = max(len(pk), 1)
        return {"quiet": float((pk <= 26843).sum() / n), "mid": float(((pk > 26843) & (pk <= 70171)).sum() / n),
                "loud": float((pk > 70171).sum() / n), "frames": int(len(pk))}

    def get_dis(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.int16)
        b = np.ascontiguousarray(b, dtype=np.int16)
        return self.L.sr_oracle_get_dis(_p(a), _p(b), C.c_uint32(len(a)))

    def dtw(self, a, na, b, nb):
        """a, b: int16 arrays [>=n+1 frames, n_coef] (one frame of slack is read by the do-while)."""
        a = np.ascontiguousarray(a, dtype=np.int16)
        b = np.ascontiguousarray(b, dtype=np.int16)
        return self.L.sr_oracle_dtw(_p(a), C.c_uint32(na), _p(b), C.c_uint32(nb), C.c_uint32(self.n_coef))

    def get_mdl(self, a, na, b, nb, out_rows):
        """DTW.C:217-296.  a, b as for dtw().  Returns (dis, merged frame count, int16 [min(count, out_rows), n_coef])."""
        a = np.ascontiguousarray(a, dtype=np.int16)
        b = np.ascontiguousarray(b, dtype=np.int16)
        out = np.zeros((max(out_rows, 1), self.n_coef), dtype=np.int16)
        nf = C.c_uint32(0)
        self.L.sr_oracle_get_mdl.restype = C.c_uint32
        dis = self.L.sr_oracle_get_mdl(_p(a), C.c_uint32(na), _p(b), C.c_uint32(nb), C.c_uint32(self.n_coef), _p(out),
                                       C.c_uint32(out_rows), C.byref(nf))
        return dis, nf.value, out[:min(nf.value, out_rows)].copy()

    def make_templates(self, tpl_mfcc, tpl_frames, tpl_valid=None):
        """tpl_mfcc: int16 [K, Tt, n_coef] (Tt >= max frames + 1)."""
        tpl_mfcc = np.ascontiguousarray(tpl_mfcc, dtype=np.int16)
        tpl_frames = np.ascontiguousarray(tpl_frames, dtype=np.uint32)
        K = tpl_mfcc.shape[0]
        if tpl_valid is None:
            tpl_valid = np.ones(K, dtype=np.uint8)
        tpl_valid = np.ascontiguousarray(tpl_valid, dtype=np.uint8)
        t = Templates(_p(tpl_mfcc).value, _p(tpl_frames).value, _p(tpl_valid).value, K,
                      tpl_mfcc.shape[1] * tpl_mfcc.shape[2])
        t._keep = (tpl_mfcc, tpl_frames, tpl_valid)
        return t

    def recognize_batch(self, pcm, tpl, n_threads=1, want_mfcc=True, want_scores=True):
        """pcm: uint16 [B, S].  Returns (results[B], mfcc[B, max_frames, C] | None, scores[B, K] | None)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        res = np.zeros(B, dtype=RESULT_DTYPE)
        mf = np.zeros((B, self.max_frames, self.n_coef), dtype=np.int16) if want_mfcc else None
        sc = np.zeros((B, tpl.n), dtype=np.uint32) if want_scores else None
        self.L.sr_oracle_recognize_batch(self.h, _p(pcm), C.c_uint64(S), C.c_uint32(S), C.c_uint32(B), C.byref(tpl),
                                         _p(res), _p(mf) if want_mfcc else None, _p(sc) if want_scores else None,
                                         C.c_uint32(n_threads))
        return res, mf, sc


def _oracle_recognize_segments(self, pcm, tpl):
    """pcm uint16 [S] -> (results[max_seg], scores[max_seg, K])"""
    pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
    ms = self.cfg.max_seg
    res = np.zeros(ms, dtype=RESULT_DTYPE)
    sc = np.zeros((ms, tpl.n), dtype=np.uint32)
    self.L.sr_oracle_recognize_segments(self.h, _p(pcm), C.c_uint32(len(pcm)), C.byref(tpl), _p(res), _p(sc))
    return res, sc


Oracle.recognize_segments = _oracle_recognize_segments


def _oracle_dtw_dp(self, a, na, b, nb):
    """NON-REFERENCE extension: full-DP DTW (own definition, sr_oracle.c)."""
    a = np.ascontiguousarray(a, dtype=np.int16)
    b = np.ascontiguousarray(b, dtype=np.int16)
    self.L.sr_oracle_dtw_dp.restype = C.c_uint32
    return self.L.sr_oracle_dtw_dp(_p(a), C.c_uint32(na), _p(b), C.c_uint32(nb), C.c_uint32(self.n_coef))


Oracle.dtw_dp = _oracle_dtw_dp


def _oracle_dtw_dp_batch(self, im, inf, tm, tf, n_threads=8):
    """all B x K pairs of the NON-REFERENCE full-DP scorer, threaded inside the C library; tf[k] == 0 = invalid slot"""
    im = np.ascontiguousarray(im, dtype=np.int16)
    tm = np.ascontiguousarray(tm, dtype=np.int16)
    inf = np.ascontiguousarray(inf, dtype=np.uint32)
    tf = np.ascontiguousarray(tf, dtype=np.uint32)
    B, K = im.shape[0], tm.shape[0]
    out = np.zeros((B, K), dtype=np.uint32)
    self.L.sr_oracle_dtw_dp_batch.restype = None
    self.L.sr_oracle_dtw_dp_batch(_p(im), _p(inf), C.c_uint32(im.shape[1]), C.c_uint32(B), _p(tm), _p(tf),
                                  C.c_uint32(tm.shape[1]), C.c_uint32(K), C.c_uint32(self.n_coef), _p(out),
                                  C.c_uint32(n_threads))
    return out


Oracle.dtw_dp_batch = _oracle_dtw_dp_batch


def _oracle_delta_mfcc(self, m, n):
    """EXTENSION (own definition, sr_oracle.c): delta cepstra of one record [>= n, n_coef] -> [n, n_coef]."""
    m = np.ascontiguousarray(m, dtype=np.int16)
    out = np.zeros((n, self.n_coef), dtype=np.int16)
    self.L.sr_oracle_delta_mfcc(_p(m), C.c_uint32(n), C.c_uint32(self.n_coef), _p(out))
    return out


Oracle.delta_mfcc = _oracle_delta_mfcc


class RefLib:
    """Tier (i): the reference's own objects.  Non-reentrant (file-scope statics): single thread only."""
    FTR_BYTES = 2860  # sizeof(v_ftr_tag) at vv_frm_max = 119 (MFCC.H:18-25)
    FRM_MAX = 119
    PATH = REF_PATH

    @classmethod
    def available(cls):
        build()
        return os.path.exists(cls.PATH)

    def __init__(self):
        build()
        self.L = C.CDLL(self.PATH)
        self.L.dtw.restype = C.c_uint32
        self.L.get_dis.restype = C.c_uint32
        self.L.dtw_limit.restype = C.c_uint8
        assert self.L.sr_ref_sizeof_ftr() == self.FTR_BYTES and self.L.sr_ref_vv_frm_max() == self.FRM_MAX

    def vad(self, pcm, noise_len=2400):
        """pcm: uint16 [S]; returns (Atap, seg[6])."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        a = Atap()
        seg = np.zeros(6, dtype=np.int32)
        self.L.sr_ref_vad(_p(pcm), C.c_uint16(len(pcm)), C.c_uint16(noise_len), C.byref(a), _p(seg))
        return a, seg

    def mfcc(self, pcm, start, end, atap):
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        ftr = np.zeros(self.FTR_BYTES, dtype=np.uint8)
        self.L.sr_ref_mfcc(_p(pcm), C.c_int32(start), C.c_int32(end), C.byref(atap), _p(ftr))
        h = ftr.view(np.int16)
        n = int(ftr.view(np.uint16)[1])
        return n, h[2:2 + n * 12].reshape(n, 12).copy(), ftr

    def fft(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        out = np.zeros(1024, dtype=np.uint32)
        self.L.cr4_fft_1024_stm32(_p(out), _p(words), C.c_uint16(1024))
        return out

    @classmethod
    def make_ftr(cls, mfcc, n, save_sign=12345):
        """Pack [n,12] int16 into a v_ftr_tag image (MFCC.H:18-25)."""
        ftr = np.zeros(cls.FTR_BYTES, dtype=np.uint8)
        ftr.view(np.uint16)[0] = save_sign
        ftr.view(np.uint16)[1] = n
        m = np.ascontiguousarray(mfcc, dtype=np.int16).reshape(-1)[:cls.FRM_MAX * 12]
        ftr.view(np.int16)[2:2 + len(m)] = m
        return ftr

    def dtw(self, ftr_in, ftr_mdl):
        return self.L.dtw(_p(ftr_in), _p(ftr_mdl))

    def get_mdl(self, ftr_in1, ftr_in2):
        """the reference's own get_mdl (DTW.C:217-296).  The output record is over-allocated to 2*119 frames because
        the reference writes `step` frames, which can exceed the 119-frame v_ftr_tag.  Returns (dis, step, rows)."""
        self.L.get_mdl.restype = C.c_uint32
        out = np.zeros(4 + 24 * 240, dtype=np.uint8)
        dis = self.L.get_mdl(_p(ftr_in1), _p(ftr_in2), _p(out))
        n = int(out.view(np.uint16)[1])
        return dis, n, out[4:].view(np.int16)[:n * 12].reshape(n, 12).copy()

    def spch_recg(self, pcm, store, stride=4096, noise_len=2400, seg_idx=0):
        """store: uint8 [n_slots*stride] flash-style image.  Returns (status, best_slot, dis, scores, mfcc, n)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        n_slots = len(store) // stride
        ftr = np.zeros(self.FTR_BYTES, dtype=np.uint8)
        best = C.c_uint32(0)
        dis = C.c_uint32(0)
        scores = np.zeros(n_slots, dtype=np.uint32)
        st = self.L.sr_ref_spch_recg_seg(_p(pcm), C.c_uint16(len(pcm)), C.c_uint16(noise_len), _p(store),
                                         C.c_uint32(n_slots), C.c_uint32(stride), C.c_uint32(seg_idx), _p(ftr),
                                         C.byref(best), C.byref(dis), _p(scores))
        n = int(ftr.view(np.uint16)[1])
        return st, best.value, dis.value, scores, ftr.view(np.int16)[2:2 + n * 12].reshape(n, 12).copy(), n


class RefLib320(RefLib):
    """Tier (i) at vv_frm_max = 320: the reference's VAD.C / MFCC.C / DTW.C compiled from where they lie with one
    compile-time constant changed (vv_tim_max 1200 -> 3210 ms, MFCC.H:15, through a sed-patched temporary copy of that
    header in a temporary build directory outside the repository; see oracle/Makefile)."""
    FTR_BYTES = 4 + 320 * 12 * 2
    FRM_MAX = 320
    PATH = REF320_PATH


def ref320_store(tm, tfr, valid=None, stride=8192):
    """v_ftr_tag images (save_sign | frm_num | mfcc_dat, MFCC.H:18-25) of K templates at byte stride `stride`, as the
    firmware's save_mdl leaves them in flash (main.c:121-138, Flash.H:11-20); valid[k] == 0 -> erased slot (0xFF bytes)"""
    K = len(tfr)
    store = np.full(K * stride, 0xFF, dtype=np.uint8)
    for k in range(K):
        if valid is not None and not valid[k]:
            continue
        rec = store[k * stride:(k + 1) * stride]
        rec[:4].view(np.uint16)[:] = (12345, tfr[k])
        rec[4:4 + int(tfr[k]) * 24] = np.ascontiguousarray(tm[k, :tfr[k]]).view(np.uint8).reshape(-1)
    return store


class Ref320Pool:
    """The reference's OWN objects at the benchmark shape, many utterances in parallel.  VAD.C / MFCC.C / DTW.C keep
    file-scope statics (MFCC.C:14-15, DTW.C:65-68), so every host thread dlopens its own private copy of
    oracle/_ref/libsr_ref320.so (copies in a temporary directory, removed by close()); ctypes drops the GIL during the calls."""

    def __init__(self, n_threads):
        import shutil
        import tempfile
        build()
        self.n = max(1, int(n_threads))
        self.tmp = tempfile.mkdtemp(prefix="sr_ref_")
        self.libs = []
        for i in range(self.n):
            pth = os.path.join(self.tmp, f"libsr_ref320_{i}.so")
            shutil.copyfile(REF320_PATH, pth)
            self.libs.append(C.CDLL(pth))
        self._rm = shutil.rmtree

    def close(self):
        if self.tmp:
            self._rm(self.tmp, ignore_errors=True)
            self.tmp = None

    def recognize(self, host, store, n_slots, stride=8192, noise_len=2400, want_mfcc=False, n_run=None, threads=None):
        """spch_recg (main.c:249-296, restated in oracle/ref_glue.c over an explicit store) on host[b], b < n_run.
        Returns dict(status, best, dis, scores[n, K], seconds[, frm_num, mfcc[n, 320, 12]])."""
        import threading
        import time
        host = np.ascontiguousarray(host, dtype=np.uint16)
        n = host.shape[0] if n_run is None else n_run
        S = host.shape[1]
        nt = self.n if threads is None else max(1, min(threads, self.n))
        r = dict(status=np.zeros(n, np.int32), best=np.zeros(n, np.uint32), dis=np.zeros(n, np.uint32),
                 scores=np.zeros((n, n_slots), np.uint32))
        if want_mfcc:
            r["frm_num"] = np.zeros(n, np.uint32)
            r["mfcc"] = np.zeros((n, RefLib320.FRM_MAX, 12), np.int16)

        def work(i, lo, hi):
            L = self.libs[i]
            ftr = np.zeros(RefLib320.FTR_BYTES, dtype=np.uint8)
            best, dis = C.c_uint32(0), C.c_uint32(0)
            for b in range(lo, hi):
                ftr[:4] = 0
                r["status"][b] = L.sr_ref_spch_recg_seg(_p(host[b]), C.c_uint16(S), C.c_uint16(noise_len), _p(store),
                                                        C.c_uint32(n_slots), C.c_uint32(stride), C.c_uint32(0), _p(ftr),
                                                        C.byref(best), C.byref(dis), _p(r["scores"][b]))
                r["best"][b], r["dis"][b] = best.value, dis.value
                if want_mfcc and r["status"][b] != 1:
                    nf = int(ftr.view(np.uint16)[1])
                    r["frm_num"][b] = nf
                    r["mfcc"][b, :nf] = ftr.view(np.int16)[2:2 + nf * 12].reshape(nf, 12)

        per = (n + nt - 1) // nt
        th = [threading.Thread(target=work, args=(i, min(i * per, n), min((i + 1) * per, n))) for i in range(nt)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        r["seconds"] = time.perf_counter() - t0
        return r
