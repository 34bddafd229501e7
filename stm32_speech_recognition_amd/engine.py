"""Host-side mirror of include/sr_engine.h (ctypes over the C ABI of libsr_engine.so).

The shared library holds the hand-written gfx950 kernels; this module only moves pointers.
PyTorch is used for device memory and streams (plumbing).  There is no fallback: if the library is
missing, or no MI355X is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsr_engine.so")
# the same sources built with -DSR_TESTING (csrc/Makefile): the development / test hooks of sr_dev_hook exist only there
TESTING_LIB_PATH = os.path.join(_HERE, "libsr_engine_testing.so")

DIS_ERR = 0xFFFFFFFF
ST_OK, ST_VAD_FAIL, ST_MFCC_FAIL, ST_SEG_OOB = 0, 1, 2, 3
N_COEF = 12

RESULT_DTYPE = np.dtype([("best_tpl", "<u4"), ("min_dis", "<u4"), ("frm_num", "<u4"), ("status", "<u4")])
VAD_DTYPE = np.dtype([("mid_val", "<u4"), ("n_thl", "<u2"), ("z_thl", "<u2"), ("s_thl", "<u4"),
                      ("seg", "<i4", (6,)), ("frm_num", "<u4"), ("status", "<u4"), ("_pad", "<u4")])
assert RESULT_DTYPE.itemsize == 16 and VAD_DTYPE.itemsize == 48


class Config(C.Structure):
    _fields_ = [("fs", C.c_uint32), ("frame_time_ms", C.c_uint32), ("frame_mov_ms", C.c_uint32),
                ("nfft", C.c_uint32), ("n_mel", C.c_uint32), ("n_coef", C.c_uint32), ("max_frames", C.c_uint32),
                ("noise_len_ms", C.c_uint32), ("max_seg", C.c_uint32), ("device", C.c_int32)]


class SrError(RuntimeError):
    pass


_lib = None
_lib_testing = None


def _open(path):
    try:  # a process that also uses PyTorch must let torch load ITS HIP runtime first: libsr_engine.so then binds to the
        import torch  # noqa: F401  same libamdhip64 (loaded the other way round, torch finds "no HIP GPUs")
    except ImportError:
        pass
    if not os.path.exists(path):
        raise SrError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    L.sr_last_error.restype = C.c_char_p
    L.sr_num_templates.restype = C.c_uint32
    L.sr_num_templates.argtypes = [C.c_void_p]
    L.sr_destroy.argtypes = [C.c_void_p]
    return L


def load_library(testing=False):
    """dlopen libsr_engine.so (built by __graft_entry__.build() / csrc/Makefile).  testing=True: the -DSR_TESTING build
    (libsr_engine_testing.so: development hooks compiled in), a second, independent library instance; the environment
    variable SR_ENGINE_TESTING=1 makes it the default of a whole process (test subprocesses that need hooks everywhere)."""
    global _lib, _lib_testing
    if testing or os.environ.get("SR_ENGINE_TESTING") == "1":
        if _lib_testing is None:
            _lib_testing = _open(TESTING_LIB_PATH)
            assert _lib_testing.sr_testing_build() == 1
        return _lib_testing
    if _lib is None:
        _lib = _open(os.environ.get("SR_ENGINE_LIB", LIB_PATH))  # development override: A/B-ing two builds on one GPU box
    return _lib


def _vp(x):
    """void* of a numpy array, a torch tensor, an int address or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


def dev_hook(name, value):
    """sr_dev_hook: development / test knobs (process-global, 0 = default); see include/sr_engine.h.  They exist only in
    the -DSR_TESTING build, so this always addresses libsr_engine_testing.so: engines that should see a hook are created
    with Engine(..., testing=True) / MultiEngine(..., testing=True) / Engine.clone(testing=True)."""
    L = load_library(testing=True)
    rc = L.sr_dev_hook(name.encode(), C.c_int64(value))
    if rc != 0:
        raise SrError(f"sr_dev_hook error {rc}: {L.sr_last_error().decode()}")


class _Tables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("hamm", "tri_cen", "tri_even", "tri_odd", "dct", "tw_kr", "tw_ki", "log_thr")]


def build_tables(**kw):
    """Host-only sr_build_tables(): the constant tables sr_create generates for a configuration (no GPU touched).
    Keywords override sr_default_config fields (fs=16000, nfft=512, n_mel=40 for the extension front end)."""
    L = load_library()
    cfg = Config()
    L.sr_default_config(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    frame_len, nb = cfg.fs // 1000 * cfg.frame_time_ms, cfg.nfft // 2
    out = dict(hamm=np.zeros(frame_len, np.uint16), tri_cen=np.zeros(cfg.n_mel, np.uint16),
               tri_even=np.zeros(nb, np.uint16), tri_odd=np.zeros(nb, np.uint16),
               dct=np.zeros(cfg.n_coef * cfg.n_mel, np.int8), tw_kr=np.zeros(1020, np.int16),
               tw_ki=np.zeros(1020, np.int16), log_thr=np.zeros(2220, np.uint32))
    t = _Tables(**{k: v.ctypes.data_as(C.c_void_p) for k, v in out.items()})
    rc = L.sr_build_tables(C.byref(cfg), C.byref(t))
    if rc != 0:
        raise SrError(f"sr_build_tables error {rc}: {L.sr_last_error().decode()}")
    out["tie_delta"] = np.zeros(32768, np.int8)  # front-end independent: its own entry point
    rc = L.sr_build_tie_table(out["tie_delta"].ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise SrError(f"sr_build_tie_table error {rc}: {L.sr_last_error().decode()}")
    return out


class Engine:
    """One sr_engine handle.  Defaults are the firmware's constants except where overridden."""

    def __init__(self, max_frames=119, device=-1, testing=False, **kw):
        self.L = load_library(testing)
        self._testing, self._kw, self._tpl = testing, dict(kw), None
        cfg = Config()
        self.L.sr_default_config(C.byref(cfg))
        cfg.max_frames = max_frames
        cfg.device = device
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        self._check(self.L.sr_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.max_frames = max_frames
        self.n_coef = cfg.n_coef  # 12 except for the generic front end
        self.noise_len = (cfg.fs // 1000) * cfg.noise_len_ms

    @classmethod
    def borrowed(cls, handle, cfg, max_frames, lib=None):
        """a view of an sr_engine handle somebody else owns (sr_multi_engine): same methods, never destroyed from here"""
        e = cls.__new__(cls)
        e.L, e.cfg, e.h, e.max_frames, e._borrowed = lib if lib is not None else load_library(), cfg, C.c_void_p(handle), max_frames, True
        e._testing, e._kw, e._tpl = False, {}, None
        e.n_coef = cfg.n_coef
        e.noise_len = (cfg.fs // 1000) * cfg.noise_len_ms
        return e

    def clone(self, testing=False):
        """a second engine with this one's configuration and template store, on the product library or (testing=True) on
        the -DSR_TESTING build -- how a test points a development hook at the shapes of an engine it already has"""
        e = Engine(max_frames=self.max_frames, device=self.cfg.device, testing=testing, **self._kw)
        if self._tpl is not None:
            getattr(e, self._tpl[0])(*self._tpl[1])
        return e

    def _check(self, rc):
        if rc != 0:
            raise SrError(f"sr_engine error {rc}: {self.L.sr_last_error().decode()}")

    def mag_cheap_bound(self):
        """sr_mag_cheap_bound: 70171 when sr_create's sweep of this device confirmed the frame kernel's cheap magnitude form, else 0"""
        self.L.sr_mag_cheap_bound.restype = C.c_uint32
        return int(self.L.sr_mag_cheap_bound(self.h))

    def lds_poison(self, seed, stream=None):
        """sr_lds_poison: overwrite every CU's local data share with a seeded pattern; returns the bytes filled per workgroup"""
        n = C.c_uint32(0)
        self._check(self.L.sr_lds_poison(self.h, C.c_uint32(seed), C.c_void_p(stream), C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.L.sr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- template store ---------------------------------------------------------------------------
    @property
    def n_templates(self):
        return self.L.sr_num_templates(self.h)

    def set_templates_dense(self, mfcc, frames, valid=None):
        """mfcc int16 [K, rows, 12]; frames uint32 [K]; valid uint8 [K] or None (all valid)."""
        mfcc = np.ascontiguousarray(mfcc, dtype=np.int16)
        frames = np.ascontiguousarray(frames, dtype=np.uint32)
        if valid is not None:
            valid = np.ascontiguousarray(valid, dtype=np.uint8)
        K = mfcc.shape[0]
        self._check(self.L.sr_set_templates_dense(self.h, _vp(mfcc), _vp(frames), _vp(valid), C.c_uint32(K),
                                                  C.c_uint32(mfcc.shape[1] * mfcc.shape[2])))
        self._tpl = ("set_templates_dense", (mfcc, frames, valid))

    def set_templates_store(self, store, stride=4096):
        """Firmware flash image: v_ftr_tag slots at `stride` bytes (Flash.H:11-20)."""
        store = np.ascontiguousarray(store, dtype=np.uint8)
        self._check(self.L.sr_set_templates(self.h, _vp(store), C.c_uint32(len(store) // stride), C.c_uint32(stride)))
        self._tpl = ("set_templates_store", (store, stride))

    def train_store(self, pcm, slots, store=None, n_slots=80, stride=4096):
        """save_mdl for each row of pcm into slot slots[i] of a flash-style store image (created erased,
        all 0xFF, when not given).  Returns (store uint8 [n_slots*stride], status uint32 [n])."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        n, S = pcm.shape
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        if store is None:
            store = np.full(n_slots * stride, 0xFF, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint32)
        self._check(self.L.sr_train_store(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S), C.c_uint32(n), _vp(slots),
                                          _vp(store), C.c_uint32(len(store) // stride), C.c_uint32(stride), _vp(status)))
        return store, status

    # ---- host-buffer API --------------------------------------------------------------------------
    def recognize(self, pcm, want_scores=True, want_mfcc=True, want_vad=True, buf_len=None):
        """pcm uint16 [B, S] on the host.  Returns dict(results, scores, mfcc, vad) of numpy arrays."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        buf_len = S if buf_len is None else buf_len
        K = self.n_templates
        res = np.zeros(B, dtype=RESULT_DTYPE)
        sc = np.zeros((B, K), dtype=np.uint32) if want_scores else None
        mf = np.zeros((B, self.max_frames, self.n_coef), dtype=np.int16) if want_mfcc else None
        vd = np.zeros(B, dtype=VAD_DTYPE) if want_vad else None
        self._check(self.L.sr_recognize_batch(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(buf_len), C.c_uint32(B),
                                              _vp(res), _vp(sc), _vp(mf), _vp(vd)))
        return dict(results=res, scores=sc, mfcc=mf, vad=vd)

    def recognize_packed12(self, packed, buf_len, want_scores=True, want_mfcc=True, want_vad=True):
        """packed uint8 [B, row_bytes]: 12-bit codes, two samples in three bytes (pack12()); sr_recognize_batch_packed12."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        B, rb = packed.shape
        K = self.n_templates
        res = np.zeros(B, dtype=RESULT_DTYPE)
        sc = np.zeros((B, K), dtype=np.uint32) if want_scores else None
        mf = np.zeros((B, self.max_frames, self.n_coef), dtype=np.int16) if want_mfcc else None
        vd = np.zeros(B, dtype=VAD_DTYPE) if want_vad else None
        self._check(self.L.sr_recognize_batch_packed12(self.h, _vp(packed), C.c_uint64(rb), C.c_uint32(buf_len), C.c_uint32(B),
                                                       _vp(res), _vp(sc), _vp(mf), _vp(vd)))
        return dict(results=res, scores=sc, mfcc=mf, vad=vd)

    def recognize_segments(self, pcm):
        """All VAD segments (extension of main.c:268).  Returns (results [max_seg, B], scores [max_seg, B, K], vad [B])."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        ms, K = self.cfg.max_seg, self.n_templates
        res = np.zeros((ms, B), dtype=RESULT_DTYPE)
        sc = np.zeros((ms, B, K), dtype=np.uint32)
        vd = np.zeros(B, dtype=VAD_DTYPE)
        self._check(self.L.sr_recognize_segments_batch(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S), C.c_uint32(B),
                                                       _vp(res), _vp(sc), _vp(vd)))
        return res, sc, vd

    def vad(self, pcm, buf_len=None):
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        vd = np.zeros(B, dtype=VAD_DTYPE)
        self._check(self.L.sr_vad_batch(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S if buf_len is None else buf_len),
                                        C.c_uint32(B), _vp(vd)))
        return vd

    def mfcc(self, pcm, start, end, mid):
        """get_mfcc of segment [start[b], end[b]) of row b with mid value mid[b]."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        mid = np.ascontiguousarray(mid, dtype=np.uint32)
        out = np.zeros((B, self.max_frames, self.n_coef), dtype=np.int16)
        n = np.zeros(B, dtype=np.uint32)
        self._check(self.L.sr_mfcc_batch(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S), C.c_uint32(B), _vp(start),
                                         _vp(end), _vp(mid), _vp(out), _vp(n)))
        return n, out

    def mfcc_status(self, pcm, start, end, mid):
        """like mfcc(), with per-record failure instead of a batch error (sr_mfcc_batch_status): returns
        (frm_num, mfcc, status); a bad record has frm_num 0, an all-zero MFCC record and status != 0."""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        mid = np.ascontiguousarray(mid, dtype=np.uint32)
        out = np.zeros((B, self.max_frames, self.n_coef), dtype=np.int16)
        n = np.zeros(B, dtype=np.uint32)
        st = np.zeros(B, dtype=np.uint32)
        self._check(self.L.sr_mfcc_batch_status(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S), C.c_uint32(B), _vp(start),
                                                _vp(end), _vp(mid), _vp(out), _vp(n), _vp(st)))
        return n, out, st

    def dtw(self, in_mfcc, in_frames):
        """in_mfcc int16 [B, max_frames, 12] against the template store -> (scores [B,K], results [B])."""
        in_mfcc = np.ascontiguousarray(in_mfcc, dtype=np.int16)
        assert in_mfcc.shape[1:] == (self.max_frames, self.n_coef)
        in_frames = np.ascontiguousarray(in_frames, dtype=np.uint32)
        B = in_mfcc.shape[0]
        sc = np.zeros((B, self.n_templates), dtype=np.uint32)
        res = np.zeros(B, dtype=RESULT_DTYPE)
        self._check(self.L.sr_dtw_batch(self.h, _vp(in_mfcc), _vp(in_frames), C.c_uint32(B), _vp(sc), _vp(res)))
        return sc, res

    def dtw_dp(self, in_mfcc, in_frames):
        """OPT-IN non-reference scorer: full-DP DTW scores [B, K] (see sr_dtw_dp_batch)."""
        in_mfcc = np.ascontiguousarray(in_mfcc, dtype=np.int16)
        assert in_mfcc.shape[1:] == (self.max_frames, self.n_coef)
        in_frames = np.ascontiguousarray(in_frames, dtype=np.uint32)
        B = in_mfcc.shape[0]
        sc = np.zeros((B, self.n_templates), dtype=np.uint32)
        self._check(self.L.sr_dtw_dp_batch(self.h, _vp(in_mfcc), _vp(in_frames), C.c_uint32(B), _vp(sc)))
        return sc

    def set_dp_lanes(self, lanes=0):
        """lanes per pair of the opt-in full-DP scorer: 0 default (8), 4 / 8 / 16 band kernel, 1 = one wave per pair"""
        self._check(self.L.sr_set_dp_lanes(self.h, C.c_uint32(lanes)))

    def set_small_launch(self, mode=0):
        """Small-launch forms of the kernels (four waves per capture in VAD, 4-frame workgroups in the frame kernel, one
        workgroup per DTW pair + in-kernel slot scan, four lanes per DTW pair for mid-sized launches, pinned host staging):
        0 automatic by launch size, 1 never, 2 always (DTW: one workgroup per pair whenever the in x mdl rectangle fits a
        workgroup's LDS), 3 the four-lanes-per-pair DTW form whenever the sequences fit.  Same results in every mode."""
        self._check(self.L.sr_set_small_launch(self.h, C.c_int(mode)))

    def dtw_dp_dev(self, mfcc, scores, in_frames=None, vad=None, stream=None):
        """OPT-IN non-reference scorer on device tensors: mfcc int16 [B, max_frames, 12], frame counts from in_frames
        (int32 [B]) or vad records; scores int32 [B, K].  Asynchronous on `stream`."""
        import torch
        B = mfcc.shape[0]
        if stream is None:
            stream = torch.cuda.current_stream(mfcc.device).cuda_stream
        self._check(self.L.sr_dtw_dp_batch_dev(self.h, _vp(mfcc), _vp(in_frames), _vp(vad), C.c_uint32(B), _vp(scores),
                                               C.c_void_p(stream)))
        return scores

    def get_mdl(self, in1, n1, in2, n2, mdl_rows):
        """get_mdl (DTW.C:217-296) on P pairs: in1 int16 [P, rows1, 12], in2 int16 [P, rows2, 12].
        Returns (mdl int16 [P, mdl_rows, 12], mdl_frames uint32 [P], dis uint32 [P])."""
        in1 = np.ascontiguousarray(in1, dtype=np.int16)
        in2 = np.ascontiguousarray(in2, dtype=np.int16)
        n1 = np.ascontiguousarray(n1, dtype=np.uint32)
        n2 = np.ascontiguousarray(n2, dtype=np.uint32)
        P = in1.shape[0]
        assert in1.shape[2] == N_COEF and in2.shape[2] == N_COEF and in2.shape[0] == P
        mdl = np.zeros((P, mdl_rows, N_COEF), dtype=np.int16)
        frames = np.zeros(P, dtype=np.uint32)
        dis = np.zeros(P, dtype=np.uint32)
        self._check(self.L.sr_get_mdl_batch(self.h, _vp(in1), _vp(n1), C.c_uint32(in1.shape[1]), _vp(in2), _vp(n2),
                                            C.c_uint32(in2.shape[1]), C.c_uint32(P), _vp(mdl), C.c_uint32(mdl_rows),
                                            _vp(frames), _vp(dis)))
        return mdl, frames, dis

    def delta_mfcc(self, mfcc, frames):
        """EXTENSION (no reference counterpart): two-frame regression delta cepstra of B records [B, max_frames, 12]."""
        mfcc = np.ascontiguousarray(mfcc, dtype=np.int16)
        frames = np.ascontiguousarray(frames, dtype=np.uint32)
        assert mfcc.shape[1:] == (self.max_frames, self.n_coef)
        out = np.zeros_like(mfcc)
        self._check(self.L.sr_delta_mfcc_batch(self.h, _vp(mfcc), _vp(frames), C.c_uint32(len(frames)), _vp(out)))
        return out

    def fft_q15(self, words):
        """cr4_fft_1024_stm32 on uint32 [n, 1024] packed complex arrays."""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        out = np.zeros_like(words)
        self._check(self.L.sr_fft_q15_batch(self.h, _vp(words), _vp(out), C.c_uint32(words.shape[0])))
        return out

    # ---- device-resident API (torch tensors as HBM handles) ---------------------------------------------
    def alloc_outputs(self, B, device, scores=True, mfcc=True, vad=True):
        import torch
        K = self.n_templates
        o = dict(results=torch.empty(B, 4, dtype=torch.int32, device=device))
        o["scores"] = torch.empty(B, K, dtype=torch.int32, device=device) if scores else None
        o["mfcc"] = torch.empty(B, self.max_frames, self.n_coef, dtype=torch.int16, device=device) if mfcc else None
        o["vad"] = torch.empty(B, 12, dtype=torch.int32, device=device) if vad else None
        return o

    def recognize_dev(self, pcm, out, buf_len=None, stream=None):
        """pcm: torch int16 [B, S] in HBM holding the u16 ADC codes; out: alloc_outputs().  Asynchronous."""
        import torch
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
        B, S = pcm.shape
        if stream is None:
            stream = torch.cuda.current_stream(pcm.device).cuda_stream
        self._check(self.L.sr_recognize_batch_dev(
            self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S if buf_len is None else buf_len), C.c_uint32(B),
            _vp(out["results"]), _vp(out["scores"]), _vp(out["mfcc"]), _vp(out["vad"]), C.c_void_p(stream)))
        return out

    def features_dev(self, pcm, buf_len=None, stream=None):
        """VAD + MFCC only (template creation = the same front end, main.c:121-138).
        Returns (vad [B,12] int32, mfcc [B,max_frames,12] int16) device tensors."""
        import torch
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
        B, S = pcm.shape
        if stream is None:
            stream = torch.cuda.current_stream(pcm.device).cuda_stream
        vad = torch.empty(B, 12, dtype=torch.int32, device=pcm.device)
        mfcc = torch.empty(B, self.max_frames, self.n_coef, dtype=torch.int16, device=pcm.device)
        self._check(self.L.sr_vad_batch_dev(self.h, _vp(pcm), C.c_uint64(S),
                                            C.c_uint32(S if buf_len is None else buf_len), C.c_uint32(B), _vp(vad),
                                            C.c_void_p(stream)))
        self._check(self.L.sr_mfcc_batch_dev(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(B), _vp(vad), _vp(mfcc),
                                             C.c_void_p(stream)))
        return vad, mfcc

    def set_pipeline(self, streams=3, min_chunk=4096, max_chunks=12):
        """chunking of recognize_dev over the engine's internal streams (streams=1: one chunk, caller's stream)"""
        self._check(self.L.sr_set_pipeline(self.h, C.c_uint32(streams), C.c_uint32(min_chunk), C.c_uint32(max_chunks)))

    def set_profiling(self, on=True):
        self._check(self.L.sr_set_profiling(self.h, C.c_int(1 if on else 0)))

    def stage_ms(self):
        ms = (C.c_float * 5)()
        self._check(self.L.sr_get_stage_ms(self.h, ms))
        n = C.c_uint32(0)
        self._check(self.L.sr_get_stage_launches(self.h, C.byref(n)))
        return dict(vad=ms[0], mfcc=ms[1], dtw=ms[2], argmin=ms[3], total=ms[4], launches_per_call=n.value)


def pack12(pcm):
    """uint16 [B, S] 12-bit ADC codes -> uint8 [B, ceil(S / 2) * 3]: sample 2i = b[3i] | (b[3i+1] & 0x0F) << 8,
    sample 2i+1 = b[3i+1] >> 4 | b[3i+2] << 4 (the layout sr_recognize_batch_packed12 reads)"""
    pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
    assert pcm.max(initial=0) < 4096, "12-bit codes only"
    B, S = pcm.shape
    if S & 1:
        pcm = np.concatenate([pcm, np.zeros((B, 1), np.uint16)], 1)
    a, b = pcm[:, 0::2].astype(np.uint32), pcm[:, 1::2].astype(np.uint32)
    out = np.empty((B, a.shape[1], 3), np.uint8)
    out[:, :, 0] = a & 0xFF
    out[:, :, 1] = (a >> 8) | ((b & 0xF) << 4)
    out[:, :, 2] = b >> 4
    return out.reshape(B, -1)


def results_from_torch(t):
    """[B,4] int32 device tensor -> numpy structured sr_result array."""
    return t.cpu().numpy().view(np.uint32).reshape(-1, 4).copy().view(RESULT_DTYPE).reshape(-1)


def vad_from_torch(t):
    return t.cpu().numpy().view(np.uint8).reshape(-1, 48).copy().view(VAD_DTYPE).reshape(-1)


class MultiEngine:
    """sr_multi: ONE process driving several MI355X (utterances sharded, templates replicated, one RCCL all-gather of
    the score matrix).  Mirrors the sr_multi_* section of include/sr_engine.h."""

    def __init__(self, devices, max_frames=119, testing=False, **kw):
        self.L = load_library(testing)
        self.L.sr_multi_num_devices.restype = C.c_uint32
        cfg = Config()
        self.L.sr_default_config(C.byref(cfg))
        cfg.max_frames = max_frames
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg, self.max_frames = cfg, max_frames
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        self._check(self.L.sr_multi_create(C.byref(cfg), devs, C.c_uint32(len(devices)), C.byref(h)))
        self.h, self.devices, self.K = h, list(devices), 0

    def _check(self, rc):
        if rc != 0:
            raise SrError(f"sr_multi error {rc}: {self.L.sr_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.sr_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_templates_dense(self, mfcc, frames, valid=None):
        mfcc = np.ascontiguousarray(mfcc, dtype=np.int16)
        frames = np.ascontiguousarray(frames, dtype=np.uint32)
        K = mfcc.shape[0]
        valid = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        self._check(self.L.sr_multi_set_templates_dense(self.h, _vp(mfcc), _vp(frames), _vp(valid), C.c_uint32(K),
                                                        C.c_uint32(mfcc.shape[1] * mfcc.shape[2])))
        self.K = K

    def set_templates_store(self, store, stride=4096):
        store = np.ascontiguousarray(store, dtype=np.uint8)
        self._check(self.L.sr_multi_set_templates(self.h, _vp(store), C.c_uint32(len(store) // stride), C.c_uint32(stride)))
        self.K = len(store) // stride

    def recognize(self, pcm):
        """pcm uint16 [B, S] on the host -> (results [B], gathered scores [B, K])"""
        pcm = np.ascontiguousarray(pcm, dtype=np.uint16)
        B, S = pcm.shape
        res = np.zeros(B, dtype=RESULT_DTYPE)
        sc = np.zeros((B, self.K), dtype=np.uint32)
        self._check(self.L.sr_multi_recognize(self.h, _vp(pcm), C.c_uint64(S), C.c_uint32(S), C.c_uint32(B), _vp(res), _vp(sc)))
        return res, sc

    def engine(self, i):
        """the sr_engine of devices[i] (sr_multi_engine), as a borrowed Engine: profiling hooks, stage-level calls"""
        self.L.sr_multi_engine.restype = C.c_void_p
        h = self.L.sr_multi_engine(self.h, C.c_uint32(i))
        if not h:
            raise SrError(f"sr_multi_engine({i}): no such device")
        cfg = Config.from_buffer_copy(self.cfg)
        cfg.device = self.devices[i]
        return Engine.borrowed(h, cfg, self.max_frames, self.L)

    def recognize_dev(self, pcm_list, results_list, scores_all_list, buf_len=None, streams=None):
        """device-resident shards: torch tensors per device (pcm int16/uint16 [Bp, S], results int32 [Bp, 4],
        scores_all int32 [n_dev*Bp, K]).  streams = None: the handle's own streams, returns when they have drained;
        streams = one torch.cuda.Stream per device: asynchronous on those streams."""
        n = len(self.devices)
        Bp, S = pcm_list[0].shape
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        st = None if streams is None else (C.c_void_p * n)(*[s.cuda_stream for s in streams])
        self._check(self.L.sr_multi_recognize_dev(self.h, arr(pcm_list), C.c_uint64(pcm_list[0].stride(0)),
                                                  C.c_uint32(S if buf_len is None else buf_len), C.c_uint32(Bp),
                                                  arr(results_list), arr(scores_all_list), st))
