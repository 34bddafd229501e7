"""Python spelling of the reference's own scalar entry points (VAD.H:24-25, MFCC.H:27, DTW.H:7,
main.c:249), bound to the C symbols of the same names that libsr_engine.so exports.  Same argument
meaning and sentinel returns as the firmware; every call is a 1-item GPU dispatch.
"""
import ctypes as C

import numpy as np

from .engine import load_library

VV_FRM_MAX = 119
MFCC_NUM = 12
VCBUF_LEN = 16000
ATAP_LEN = 2400
DIS_ERR = 0xFFFFFFFF


class atap_tag(C.Structure):  # VAD.H:10-16
    _fields_ = [("mid_val", C.c_uint32), ("n_thl", C.c_uint16), ("z_thl", C.c_uint16), ("s_thl", C.c_uint32)]


class valid_tag(C.Structure):  # VAD.H:18-22
    _fields_ = [("start", C.POINTER(C.c_uint16)), ("end", C.POINTER(C.c_uint16))]


class v_ftr_tag(C.Structure):  # MFCC.H:18-25
    _pack_ = 1
    _fields_ = [("save_sign", C.c_uint16), ("frm_num", C.c_uint16), ("mfcc_dat", C.c_int16 * (VV_FRM_MAX * MFCC_NUM))]


assert C.sizeof(v_ftr_tag) == 2860


def _lib():
    L = load_library()
    L.fft.restype = C.POINTER(C.c_uint32)
    L.get_dis.restype = C.c_uint32
    L.dtw.restype = C.c_uint32
    L.get_mdl.restype = C.c_uint32
    L.dtw_limit.restype = C.c_uint8
    L.spch_recg.restype = C.c_void_p
    return L


def _addr(buf, off=0):
    return C.cast(buf.ctypes.data + 2 * off, C.POINTER(C.c_uint16))


def noise_atap(noise, n_len, atap):
    _lib().noise_atap(_addr(noise), C.c_uint16(n_len), C.byref(atap))


def VAD(vc, buf_len, atap):
    """Returns [(start, end)] * 3 as sample offsets into vc, None for NULL."""
    vv = (valid_tag * 3)()
    _lib().VAD(_addr(vc), C.c_uint16(buf_len), vv, C.byref(atap))
    base = vc.ctypes.data
    out = []
    for i in range(3):
        s = C.cast(vv[i].start, C.c_void_p).value
        e = C.cast(vv[i].end, C.c_void_p).value
        out.append((None if s is None else (s - base) // 2, None if e is None else (e - base) // 2))
    return out


def get_mfcc(vc, start, end, atap, name="get_mfcc"):
    v = valid_tag(_addr(vc, start), _addr(vc, end))
    ftr = v_ftr_tag()
    getattr(_lib(), name)(C.byref(v), C.byref(ftr), C.byref(atap))
    return ftr


def fft(frame):
    frame = np.ascontiguousarray(frame, dtype=np.int16)
    p = _lib().fft(frame.ctypes.data_as(C.POINTER(C.c_int16)), C.c_uint16(len(frame)))
    if not p:
        return None
    return np.ctypeslib.as_array(p, shape=(1024,)).copy()


def cr4_fft_1024_stm32(words):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.zeros(1024, dtype=np.uint32)
    _lib().cr4_fft_1024_stm32(out.ctypes.data_as(C.c_void_p), words.ctypes.data_as(C.c_void_p), C.c_uint16(1024))
    return out


def get_dis(a, b):
    a = np.ascontiguousarray(a, dtype=np.int16)
    b = np.ascontiguousarray(b, dtype=np.int16)
    return _lib().get_dis(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))


def dtw_limit(x, y):
    return _lib().dtw_limit(C.c_uint16(x), C.c_uint16(y))


def make_ftr(mfcc, n, save_sign=12345):
    f = v_ftr_tag()
    f.save_sign = save_sign
    f.frm_num = n
    m = np.ascontiguousarray(mfcc, dtype=np.int16).reshape(-1)[:VV_FRM_MAX * MFCC_NUM]
    C.memmove(f.mfcc_dat, m.ctypes.data, m.nbytes)
    return f


def dtw(ftr_in, ftr_mdl):
    return _lib().dtw(C.byref(ftr_in), C.byref(ftr_mdl))


def get_mdl(ftr_in1, ftr_in2):
    """DTW.C:217-296.  Returns (dis, merged v_ftr_tag)."""
    out = v_ftr_tag()
    dis = _lib().get_mdl(C.byref(ftr_in1), C.byref(ftr_in2), C.byref(out))
    return dis, out


def get_mean(a, b):
    """DTW.C:195-205 on two 12-coefficient frames."""
    a = np.ascontiguousarray(a, dtype=np.int16)
    b = np.ascontiguousarray(b, dtype=np.int16)
    m = np.zeros(MFCC_NUM, dtype=np.int16)
    _lib().get_mean(_addr(a), _addr(b), _addr(m))
    return m


def set_templates(store, stride=4096):
    store = np.ascontiguousarray(store, dtype=np.uint8)
    rc = _lib().sr_compat_set_templates(store.ctypes.data_as(C.c_void_p), C.c_uint32(len(store) // stride),
                                        C.c_uint32(stride))
    if rc:
        raise RuntimeError(_lib().sr_last_error().decode())


def spch_recg(v_dat):
    """Returns (label bytes or None, mtch_dis)."""
    v_dat = np.ascontiguousarray(v_dat, dtype=np.uint16)
    assert len(v_dat) >= VCBUF_LEN
    dis = C.c_uint32(0)
    p = _lib().spch_recg(_addr(v_dat), C.byref(dis))
    if not p:
        return None, dis.value
    return C.string_at(p), dis.value
