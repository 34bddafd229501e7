"""CPU tests that pin the oracle (no GPU).

 * tier (ii) (own restatement) against the committed golden fixtures, which were produced by
   tier (i) = the reference's own .C objects  -> runs anywhere, including the GPU box;
 * tier (ii) against tier (i) live on fresh random inputs, and the regenerated constant tables
   against the reference headers / assembly table -> only where /root/reference exists.
"""
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
from conftest import REFERENCE, needs_reference
from stm32_speech_recognition_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def oracle():
    return ol.Oracle(max_frames=119)


def _ref_text(rel):
    return open(os.path.join(REFERENCE, rel), "rb").read().decode("latin1")


# ----------------------------------------------------------------------------- tables
@needs_reference
def test_tables_regenerate_reference_headers(oracle):
    """MFCC_Arg.h:6-44 -- every constant table regenerates with 0 mismatches."""
    src = _ref_text("Src/Speech_Recog/MFCC_Arg.h")
    t = oracle.tables()
    for name, key in (("hamm", "hamm"), ("tri_cen", "tri_cen"), ("tri_odd", "tri_odd"), ("tri_even", "tri_even"),
                      ("dct_arg", "dct")):
        m = re.search(name + r"\[\]\s*=\s*\{([^}]*)\}", src)
        want = np.array([int(v) for v in re.findall(r"-?\d+", m.group(1))])
        got = t[key].astype(np.int64)
        assert len(want) == len(got), name
        assert (want != got).sum() == 0, name


@needs_reference
def test_twiddles_regenerate_asm_table():
    """cr4_fft_1024_stm32.s:285-629 -- all 1020 (Kr', Ki) entries regenerate exactly."""
    import ctypes as C
    s = _ref_text("Src/BSP/cr4_fft_1024_stm32.s")
    tab = s[s.index("TableFFT_V7", s.index("passloop_v7")):]
    tab = tab[tab.index("\n"):]
    hw = [int(h, 16) for line in tab.splitlines() if "DCW" in line
          for h in re.findall(r"0x([0-9a-fA-F]{4})", line.split(";")[0])]
    hw = np.array(hw, dtype=np.uint16).view(np.int16)
    assert len(hw) == 2040
    L = ol.Oracle().L
    kr = np.zeros(1020, dtype=np.int16)
    ki = np.zeros(1020, dtype=np.int16)
    L.sr_oracle_q15_twiddles(kr.ctypes.data_as(C.c_void_p), ki.ctypes.data_as(C.c_void_p))
    assert (hw[0::2] != kr).sum() == 0 and (hw[1::2] != ki).sum() == 0


def _asm_table():
    s = _ref_text("Src/BSP/cr4_fft_1024_stm32.s")
    tab = s[s.index("TableFFT_V7", s.index("passloop_v7")):]
    tab = tab[tab.index("\n"):]
    hw = [int(h, 16) for line in tab.splitlines() if "DCW" in line
          for h in re.findall(r"0x([0-9a-fA-F]{4})", line.split(";")[0])]
    return np.array(hw, dtype=np.uint16).view(np.int16)


@needs_reference
def test_product_tables_regenerate_reference_headers_and_asm_table():
    """The PRODUCT's generator (csrc/sr_tables.cpp through the host-only C-ABI call sr_build_tables), not the oracle's
    copy of the formulas: MFCC_Arg.h:6-44 and cr4_fft_1024_stm32.s:285-629 regenerate with 0 mismatches."""
    from stm32_speech_recognition_amd import engine
    t = engine.build_tables()
    src = _ref_text("Src/Speech_Recog/MFCC_Arg.h")
    for name, key in (("hamm", "hamm"), ("tri_cen", "tri_cen"), ("tri_odd", "tri_odd"), ("tri_even", "tri_even"),
                      ("dct_arg", "dct")):
        m = re.search(name + r"\[\]\s*=\s*\{([^}]*)\}", src)
        want = np.array([int(v) for v in re.findall(r"-?\d+", m.group(1))])
        assert len(want) == len(t[key]) and (want != t[key].astype(np.int64)).sum() == 0, name
    hw = _asm_table()
    assert len(hw) == 2040 and np.array_equal(hw[0::2], t["tw_kr"]) and np.array_equal(hw[1::2], t["tw_ki"])


def test_product_tables_equal_oracle_tables(oracle):
    """Runs everywhere (also on the GPU box, where the reference tree is absent): product generator == oracle generator
    for both front ends, and the log step table is the step function of the host's own (u32)(log((double)n)*100)."""
    import ctypes as C
    from stm32_speech_recognition_amd import engine
    for kw, orc in ((dict(), oracle), (dict(fs=16000, nfft=512, n_mel=40), ol.Oracle(max_frames=64, fs=16000, nfft=512, n_mel=40))):
        t, o = engine.build_tables(**kw), orc.tables()
        for key in ("hamm", "tri_cen", "tri_odd", "tri_even", "dct"):
            assert np.array_equal(t[key].astype(np.int64), o[key].astype(np.int64)), (kw, key)
    kr, ki = np.zeros(1020, np.int16), np.zeros(1020, np.int16)
    oracle.L.sr_oracle_q15_twiddles(kr.ctypes.data_as(C.c_void_p), ki.ctypes.data_as(C.c_void_p))
    t = engine.build_tables()
    assert np.array_equal(kr, t["tw_kr"]) and np.array_equal(ki, t["tw_ki"])
    thr = t["log_thr"].astype(np.uint64)
    assert thr[0] == 1 and thr[2219] == 0xFFFFFFFF and (np.diff(thr[:2219].astype(np.int64)) >= 0).all()
    f = lambda n: (np.log(n.astype(np.float64)) * 100).astype(np.uint64)          # numpy's log = the platform libm's
    m = np.arange(1, 2219, dtype=np.uint64)                                        # thr[m] = min{n : f(n) >= m}
    assert (f(thr[1:2219]) >= m).all() and (f(thr[1:2219] - 1) < m).all()


def test_generic_front_end_tables_and_config_gate():
    """GENERIC front end (the reference's #define constants as run-time values): the product's table generator equals the
    oracle's for every tested configuration (host-only sr_build_tables), and sr_build_tables / sr_create refuse what no
    kernel is built for with SR_ERR_BAD_CONFIG."""
    from stm32_speech_recognition_amd import engine
    for ekw, okw in ol.GENERIC_CONFIGS:
        t, o = engine.build_tables(**ekw), ol.Oracle(max_frames=64, **okw).tables()
        for key in ("hamm", "tri_cen", "tri_odd", "tri_even", "dct"):
            assert t[key].shape == o[key].shape and np.array_equal(t[key].astype(np.int64), o[key].astype(np.int64)), (ekw, key)
    for bad in (dict(nfft=512), dict(nfft=2048), dict(fs=8001), dict(fs=10000, frame_time_ms=40, frame_mov_ms=20), dict(frame_time_ms=25), dict(frame_time_ms=20, frame_mov_ms=5),
                dict(n_mel=25), dict(n_mel=66), dict(n_mel=2), dict(n_coef=0), dict(n_coef=17), dict(fs=16000, nfft=512),
                dict(fs=44000, frame_time_ms=20, frame_mov_ms=10)):
        with pytest.raises(engine.SrError, match="error 2"):
            engine.build_tables(**bad)


def test_dtw_limit_interval_form_used_by_the_dp_kernel_and_bench(oracle):
    """The full-DP kernel (and bench.py's cell count) evaluates dtw_limit (DTW.C:76-109) once per column as an interval
    lb(x) <= y <= ub(x): ub = x < X1 ? 2x+1 : ((x + 5 - in + 2 mdl) >> 1) - 1, lb = x < X2 ? x >> 1 : 2x + mdl - 2 in - 3.
    Checked point by point against the oracle's dtw_limit for every length pair up to 70 x 70 that passes the gate, and
    bench.dp_cells_per_pair against the same count."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    L = oracle.L
    n_pairs = 0
    for in_n in range(1, 71):
        for mdl_n in range(1, 71):
            if in_n > 2 * mdl_n or 2 * in_n < mdl_n:
                continue
            X1, X2 = int((2 * mdl_n - in_n) / 3) & 0xFFFF, int((4 * in_n - 2 * mdl_n) / 3) & 0xFFFF
            cells = 0
            for x in range(1, in_n + 1):
                ub = 2 * x + 1 if x < X1 else ((x + 5 - in_n + 2 * mdl_n) >> 1) - 1
                lb = (x >> 1) if x < X2 else 2 * x + mdl_n - 2 * in_n - 3
                for y in range(1, mdl_n + 1):
                    assert (lb <= y <= ub) == (L.sr_oracle_dtw_outside(x, y, in_n, mdl_n) == 0), (in_n, mdl_n, x, y)
                cells += max(0, min(ub, mdl_n) - max(lb, 1) + 1)
            # k_dtw_cells evaluates every point the walk could stand on, including the column / row one past a 1-frame
            # sequence (the do-while of DTW.C:150-188 tests x + 1 and y + 1 before the loop condition is looked at)
            for x in (in_n + 1, in_n + 2):
                ub = 2 * x + 1 if x < X1 else ((x + 5 - in_n + 2 * mdl_n) >> 1) - 1
                lb = (x >> 1) if x < X2 else 2 * x + mdl_n - 2 * in_n - 3
                for y in range(1, mdl_n + 3):
                    assert (lb <= y <= ub) == (L.sr_oracle_dtw_outside(x, y, in_n, mdl_n) == 0), (in_n, mdl_n, x, y)
            for x in range(1, in_n + 1):
                ub = 2 * x + 1 if x < X1 else ((x + 5 - in_n + 2 * mdl_n) >> 1) - 1
                lb = (x >> 1) if x < X2 else 2 * x + mdl_n - 2 * in_n - 3
                for y in (mdl_n + 1, mdl_n + 2):
                    assert (lb <= y <= ub) == (L.sr_oracle_dtw_outside(x, y, in_n, mdl_n) == 0), (in_n, mdl_n, x, y)
            got, ok = bench.dp_cells_per_pair(in_n, [mdl_n])
            assert ok == 1 and got == cells, (in_n, mdl_n, got, cells)
            n_pairs += 1
    assert n_pairs > 2000


def test_preemphasis_float_form_is_exact():
    """MFCC.C:119 multiplies the previous sample by hp_ratio = 95/100 in integer arithmetic (p*95/100, truncated toward
    zero).  The frame kernels evaluate it as (int)((float)p * 0.95000005f) -- convert, IEEE multiply, truncating convert --
    for |p| <= 65535 (u16 sample - u16 mid).  Same IEEE operations here, whole domain."""
    p = np.arange(-65535, 65536, dtype=np.int64)
    want = (np.abs(p) * 95 // 100) * np.sign(p)                               # C division truncates toward zero
    c = np.float32(0.95000005)
    assert c == np.nextafter(np.float32(0.95), np.float32(1))
    got = np.trunc((p.astype(np.float32) * c).astype(np.float32)).astype(np.int64)
    assert np.array_equal(got, want)


def test_window_fused_multiplier_is_exact():
    """Round 5: the frame kernels fold MFCC.C:122's division by hamm_top/10 into the window weight -- trunc(t*h/1000) ==
    ((t << 5) * ceil(h * 2^27 / 1000) + (t < 0 ? 2^32 - 1 : 0)) >> 32 in 64-bit signed arithmetic (one v_mad_i64_i32; csrc/sr_tables.h
    hamm_fused_multiplier, sr_dev.h window_quotient) -- and take the pre-emphasis term negated (multiply by -0.95000005f).  The
    same integer / IEEE operations here, for EVERY t the two u16 samples and the u16 mid value can produce and every window
    weight of the two specialised front ends, against C's truncating division."""
    from stm32_speech_recognition_amd.engine import build_tables
    hs = set()
    for kw in (dict(), dict(fs=16000, nfft=512, n_mel=40)):
        hs |= set(int(v) for v in build_tables(**kw)["hamm"])
    assert max(hs) == 10000 and len(hs) > 100
    t = np.arange(-131071, 131072, dtype=np.int64)       # |t| <= 65535 + 62258
    for h in sorted(hs):
        M = (h * (1 << 27) + 999) // 1000
        assert M < (1 << 31)
        u = t << 5
        P = u * M + np.where(u < 0, (1 << 32) - 1, 0).astype(np.int64)
        v = t * h
        want = np.where(v >= 0, v // 1000, -((-v) // 1000))
        assert np.array_equal(P >> 32, want), h
    p = np.arange(-65535, 65536, dtype=np.int64)
    want = np.where(p >= 0, (p * 95) // 100, -((-p * 95) // 100))
    got = np.trunc((p.astype(np.float32) * np.float32(-0.95000005)).astype(np.float32)).astype(np.int64)
    assert np.array_equal(-got, want)


def test_log_step_table_is_checked_against_the_shipped_positions():
    """MFCC.C:168 on the device is a step function whose positions sr_create finds with the HOST's libm log.  A host whose
    log differs in the last bit at one of the 2219 integer crossings would silently shift a step relative to the golden
    fixtures: the library ships the positions of the libm the fixtures were made with (csrc/sr_log_thr_ref.inc), compares,
    reports (sr_log_table_mismatches, warning in sr_last_error and on stderr) and uses the shipped ones.  Here: this host
    agrees; a perturbed host table (development hook) is detected and replaced; "log_thr_from_host" keeps the host's."""
    import subprocess
    import sys
    code = ("import ctypes as C, os\n"
            "from stm32_speech_recognition_amd.engine import build_tables, load_library, dev_hook\n"
            "dev_hook('perturb_log_thr', int(os.environ.get('T_PERTURB', '0')))\n"
            "dev_hook('log_thr_from_host', int(os.environ.get('T_FROM_HOST', '0')))\n"
            "t = build_tables()['log_thr']\n"
            "L = load_library(); L.sr_last_error.restype = C.c_char_p\n"
            "print(L.sr_log_table_mismatches(), int(t[1000]), int(t[1001]), L.sr_last_error().decode()[:40])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("T_PERTURB", "T_FROM_HOST")}
        e["SR_ENGINE_TESTING"] = "1"  # the hooks exist only in the -DSR_TESTING build: the whole subprocess binds to it
        p = subprocess.run([sys.executable, "-c", code], env=dict(e, **env), capture_output=True, text=True, cwd=root, timeout=120)
        assert p.returncode == 0, p.stderr
        f = p.stdout.split(None, 3)
        return int(f[0]), int(f[1]), int(f[2]), (f[3] if len(f) > 3 else ""), p.stderr

    bad, t1000, t1001, msg, err = run()
    assert bad == 0 and "warning" not in msg and "warning" not in err       # the test host's libm agrees with the shipped table
    bad, p1000, p1001, msg, err = run(T_PERTURB="1000")
    assert bad == 1 and (p1000, p1001) == (t1000, t1001) and msg.startswith("warning") and "libm" in err
    bad, h1000, _, _, _ = run(T_PERTURB="1000", T_FROM_HOST="1")
    assert bad == 1 and h1000 == t1000 + 1


# ----------------------------------------------------------------------------- golden (tier i outputs)
def _pack(frame_real):
    return frame_real.view(np.uint16).astype(np.uint32)


@needs_reference
def test_asm_fft_interpreted_equals_restatement(golden):
    """The reference's assembly FFT, run FROM ITS OWN SOURCE TEXT by oracle/arm_fft_interp.py (armasm / Thumb-2 subset
    interpreter; the image has no ARM toolchain), against the C restatement and against the committed golden outputs
    the GPU tests use.  This is what pins oracle/q15_fft.c -- and through the fixture the GPU FFT -- by execution of
    the reference's own instructions."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import arm_fft_interp
    m = arm_fft_interp.AsmFft(os.path.join(REFERENCE, "Src", "BSP", "cr4_fft_1024_stm32.s"))
    assert len(m.prog) > 100 and len(m.data) == 2 * 2040     # 134 instructions after macro expansion, 3 x 340 x 2 halfwords
    r = ol.RefLib()
    fin, fout = golden["fft_in"], golden["fft_out"]
    for i in range(len(fin)):                                 # every committed golden vector
        out, steps = m.run(fin[i])
        assert steps == 80336
        assert np.array_equal(np.array(out, dtype=np.uint32), fout[i]), i
    rng = np.random.default_rng(99)
    for t in range(6):                                        # fresh inputs: random at several scales, extremes, a real frame
        amp = [3, 700, 32767, 32767, 32767, 12000][t]
        re = rng.integers(-amp, amp + 1, 1024).astype(np.int16)
        im = rng.integers(-amp, amp + 1, 1024).astype(np.int16)
        if t == 3:
            re[:], im[:] = rng.choice([-32768, 32767], 1024), rng.choice([-32768, 32767], 1024)
        if t == 5:
            re[160:], im[:] = 0, 0
        w = re.view(np.uint16).astype(np.uint32) | (im.view(np.uint16).astype(np.uint32) << 16)
        out, _ = m.run(w)
        assert np.array_equal(np.array(out, dtype=np.uint32), r.fft(w)), t


def test_fft_mag_matches_golden(oracle, golden):
    """rows 16.. of the FFT fixture are zero-padded real frames, the shape fft() (MFCC.C:27-62) feeds."""
    fin, fout = golden["fft_in"], golden["fft_out"]
    for i in range(16, fin.shape[0]):
        frame = (fin[i, :160] & 0xFFFF).astype(np.uint16).view(np.int16)
        re = (fout[i, :512] & 0xFFFF).astype(np.uint16).view(np.int16).astype(np.int32)
        im = (fout[i, :512] >> 16).astype(np.uint16).view(np.int16).astype(np.int32)
        want = (np.sqrt((re * re + im * im).astype(np.float32)) * np.float32(10)).astype(np.uint32)
        assert np.array_equal(oracle.fft_mag(frame), want)


def test_vad_mfcc_match_golden(oracle, golden):
    pcm = golden["pcm"]
    for b in range(pcm.shape[0]):
        rc, a = oracle.noise_atap(pcm[b])
        assert rc == 0 and a.astuple() == tuple(golden["atap"][b])
        seg = oracle.vad(pcm[b], a)
        assert np.array_equal(seg, golden["seg"][b])
        n, m = oracle.mfcc(pcm[b], seg[0], seg[1], a)
        assert n == golden["frm_num"][b]
        assert np.array_equal(m, golden["mfcc"][b, :n])


def test_direct_mfcc_edge_cases_match_golden(oracle, golden):
    """all-zero frames (log(0)), s16 wrap after windowing, full-range u16 codes."""
    dp, dmid, dm = golden["direct_pcm"], golden["direct_mid"], golden["direct_mfcc"]
    nfd = dm.shape[1]
    for d in range(dp.shape[0]):
        a = ol.Atap(int(dmid[d]), 10, 2, 1000)
        n, m = oracle.mfcc(dp[d], 1, 1 + 160 + 80 * (nfd - 1), a)
        assert n == nfd and np.array_equal(m, dm[d])
    assert not dm[0].any() and dm[1].any()


def test_dtw_matches_golden(oracle, golden):
    ln, da, db, dd = golden["dtw_len"], golden["dtw_a"], golden["dtw_b"], golden["dtw_dis"]
    pad = np.zeros((1, 12), dtype=np.int16)
    got = np.array([oracle.dtw(np.concatenate([da[p], pad]), ln[p, 0], np.concatenate([db[p], pad]), ln[p, 1])
                    for p in range(len(dd))], dtype=np.uint32)
    assert np.array_equal(got, dd)
    assert (dd == ol.DIS_ERR).sum() > 20 and (dd != ol.DIS_ERR).sum() > 100


def test_get_mdl_matches_golden(oracle, golden):
    """template averaging (DTW.C:217-296): merged frames, frame count and distance from the reference's own get_mdl"""
    ln, da, db = golden["dtw_len"], golden["dtw_a"], golden["dtw_b"]
    md, mn, mm = golden["mdl_dis"], golden["mdl_frames"], golden["mdl_rows"]
    pad = np.zeros((1, 12), dtype=np.int16)
    for p in range(len(md)):
        dis, n, rows = oracle.get_mdl(np.concatenate([da[p], pad]), ln[p, 0], np.concatenate([db[p], pad]), ln[p, 1], 238)
        assert dis == md[p] and n == mn[p] and np.array_equal(rows, mm[p, :n]), p
        if dis != ol.DIS_ERR:
            assert dis == golden["dtw_dis"][p]            # same walk as dtw()
            assert min(ln[p]) <= n <= max(ln[p].sum() - 1, 2)   # the walk stops when either sequence ends
    assert (md != ol.DIS_ERR).sum() > 60 and (mn > 119).sum() > 10  # includes merged templates the 119-frame record cannot hold
    # clipping: only out_rows frames are stored, the count still reports the full length
    p = int(np.argmax(mn))
    dis, n, rows = oracle.get_mdl(np.concatenate([da[p], pad]), ln[p, 0], np.concatenate([db[p], pad]), ln[p, 1], 50)
    assert n == mn[p] and rows.shape[0] == 50 and np.array_equal(rows, mm[p, :50])


def test_dct_magic_multiplier_is_exact():
    """k_mfcc / k_mfcc_ext evaluate the DCT term (s32)pow * dct / 100 (MFCC.C:179, truncation toward zero) as
    sign(c) * (((pow << 14) * M_c) >> 32) with M_c = ceil(|c| * 2^18 / 100).  Exhaustive over the operand ranges:
    pow = (u32)(log(x) * 100) <= 2218 for any u32 x, c any s8."""
    pw = np.arange(0, 2219, dtype=np.int64)[:, None]
    c = np.arange(-128, 128, dtype=np.int64)[None, :]
    want = np.sign(c) * ((pw * np.abs(c)) // 100)               # C division truncates toward zero
    M = (np.abs(c) * 262144 + 99) // 100
    got = np.sign(c) * (((pw << 14) * M) >> 32)
    assert np.array_equal(got, want)
    assert int(np.log(float(2 ** 32 - 1)) * 100) == 2218 and (pw.max() << 14) < 2 ** 32 and M.max() < 2 ** 24


def _root(oracle, x):
    """g(d) = (u32)sqrtf((float)d), DTW.C:59, evaluated by the oracle's C expression"""
    import ctypes as C
    x = np.ascontiguousarray(x, dtype=np.uint32)
    out = np.zeros(3 * len(x), np.uint32)
    oracle.L.sr_oracle_math_diag(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint32(len(x)))
    return out[1::3].astype(np.int64)


def test_dtw_tie_threshold_table_is_exact(oracle):
    """k_dtw_lds takes ONE root per step, g(min of the squared candidates), and decides the reference's tie order
    (DTW.C:168-184) by comparing the other squared candidates with T(g) = first d whose root is g + 1, which the
    product builds as a byte table: T(g) = g*(g+2) + tie_delta[g], g < 32768 (csrc/sr_tables.cpp, exported by the
    host-only sr_build_tables).  g(d) = (u32)sqrtf((float)d) is monotone, so the table is exact iff
    g(T - 1) == g and g(T) == g + 1 for every entry -- checked with the C expression itself."""
    from stm32_speech_recognition_amd.engine import build_tables
    dl = build_tables()["tie_delta"].astype(np.int64)
    assert len(dl) == 32768 and (dl[:4096] == 1).all() and dl.min() >= -128
    r = np.arange(len(dl), dtype=np.int64)
    T = r * (r + 2) + dl
    assert T.min() >= 1 and T.max() < 2 ** 32
    assert np.array_equal(_root(oracle, T - 1), r)       # the largest value the kernel calls a tie really ties
    assert np.array_equal(_root(oracle, T), r + 1)       # the smallest value it calls "greater" really is greater
    # monotonicity of g itself on a dense sample around every 7th perfect square and across the 2^24 rounding knee
    M = (np.arange(0, 65535, dtype=np.int64) + 1) ** 2
    xs = np.unique(np.clip(np.concatenate([M[::7, None] + np.arange(-3, 4)[None, :],
                                           (2 ** 24 + np.arange(-2000, 2000))[None, :]], axis=None), 0, 2 ** 32 - 1))
    assert (np.diff(_root(oracle, xs)) >= 0).all()


def test_dtw_root_from_below_is_one_short_at_most(oracle):
    """k_dtw_lds takes the root of a step as floor(pred(v_sqrt_f32((float)d))) (round 6) and learns from its own tie threshold
    whether that is one short: d >= T(mn) <=> the root is mn + 1 (exact by the table test above).  What has to hold for it:
    whatever 1-ulp-accurate value s0 the device's v_sqrt_f32 returns -- the correctly rounded root r or one of its two
    neighbours -- floor(pred(s0)) is g = floor(r) or g - 1, never more and never less.  Modelled here with numpy's correctly
    rounded float32 sqrt for all three candidates of s0, on the d around every perfect square and on random d; the device
    itself is swept over all 2^32 inputs (tests/exhaustive_math_sweep.py through sr_math_diag)."""
    k = np.arange(1, 65536, dtype=np.int64)
    rng = np.random.default_rng(3)
    ds = [np.clip(k * k + off, 0, 2 ** 32 - 1) for off in list(range(-40, 41)) + [-600, -300, -129, 129, 300, 600]]
    ds.append(rng.integers(0, 2 ** 32, 2_000_000, dtype=np.int64))
    for d in ds:
        g = _root(oracle, d)
        r = np.sqrt(d.astype(np.uint32).astype(np.float32))          # correctly rounded root of (float)d
        assert np.array_equal(np.floor(r).astype(np.int64), g)
        for step in (-1, 0, 1):                                      # s0 = pred(r), r, succ(r)
            s0 = (r.view(np.int32) + step).view(np.float32)
            p = (s0.view(np.int32) - 1).view(np.float32)             # pred(s0)
            mn = np.floor(np.where(s0 > 0, p, 0)).astype(np.int64)
            ok = (mn == g) | (mn == g - 1)
            ok |= r == 0                                             # d = 0: the device converts NaN to 0 = g
            assert ok.all(), (step, d[~ok][:5], mn[~ok][:5], g[~ok][:5])


def store_to_templates(store, stride=4096, tmax=120, nc=12):
    """Firmware flash image (Flash.H:11-20, MFCC.H:18-25) -> dense batched layout."""
    K = len(store) // stride
    tm = np.zeros((K, tmax, nc), dtype=np.int16)
    tf = np.zeros(K, dtype=np.uint32)
    tv = np.zeros(K, dtype=np.uint8)
    for k in range(K):
        slot = store[k * stride:(k + 1) * stride]
        tv[k] = slot[:2].view(np.uint16)[0] == 12345
        tf[k] = slot[2:4].view(np.uint16)[0]
        body = slot[4:4 + 119 * nc * 2].view(np.int16)
        tm[k, :119] = body.reshape(119, nc)
    return tm, tf, tv


def test_recognize_matches_golden(oracle, golden):
    tm, tf, tv = store_to_templates(golden["store"])
    tpl = oracle.make_templates(tm, tf, tv)
    res, mf, sc = oracle.recognize_batch(golden["pcm"], tpl, n_threads=2)
    assert np.array_equal(res["status"], golden["recg_status"])
    assert np.array_equal(res["best_tpl"], golden["recg_best"])
    assert np.array_equal(res["min_dis"], golden["recg_dis"])
    assert np.array_equal(sc, golden["recg_scores"])
    assert (sc[:, 7] == ol.DIS_ERR).all()  # the erased slot


# ----------------------------------------------------------------------------- edge behaviour of the restatement
def test_noise_atap_bad_length_is_silent_noop(oracle):
    x = np.full(2400, 2048, dtype=np.uint16)
    rc, a = oracle.noise_atap(x, n_len=2399)  # VAD.C:33-36
    assert rc == 1 and a.astuple() == (0, 0, 0, 0)


def test_vad_failure_and_status(oracle):
    x = np.full(16000, 2048, dtype=np.uint16)  # silence: no segment
    tpl = oracle.make_templates(np.zeros((2, 120, 12), np.int16), np.array([10, 10], np.uint32))
    res, _, sc = oracle.recognize_batch(x[None], tpl)
    assert res["status"][0] == ol.ST_VAD_FAIL and res["min_dis"][0] == ol.DIS_ERR and res["best_tpl"][0] == 0
    assert (sc == ol.DIS_ERR).all()


def test_mfcc_too_long_returns_zero_frames(oracle):
    x = np.full(16000, 2048, dtype=np.uint16)
    a = ol.Atap(2048, 10, 2, 1000)
    n, _ = oracle.mfcc(x, 1, 1 + 160 + 80 * 119, a)  # 120 frames > vv_frm_max (MFCC.C:103-107)
    assert n == 0
    n, _ = oracle.mfcc(x, 1, 1 + 160 + 80 * 118, a)
    assert n == 119


def test_dtw_gates_and_identity(oracle):
    rng = np.random.default_rng(5)
    a = rng.integers(-500, 500, (121, 12)).astype(np.int16)
    assert oracle.dtw(a, 50, a, 50) == 0
    assert oracle.dtw(a, 101, a, 50) == ol.DIS_ERR  # in > 2*mdl (DTW.C:133)
    assert oracle.dtw(a, 24, a, 50) == ol.DIS_ERR   # 2*in < mdl
    assert oracle.dtw(a, 100, a, 50) != ol.DIS_ERR
    assert oracle.dtw(a, 25, a, 50) != ol.DIS_ERR


def test_synth_yields_exact_frame_count():
    """The benchmark generator must give exactly T frames through the oracle's VAD (T = 256 shape)."""
    o = ol.Oracle(max_frames=256)
    bank = synth.word_bank(10)
    T = 256
    x = synth.as_u16_numpy(synth.make_utterances(np.arange(6) % 10, [T] * 6, seed=3, bank=bank))
    assert x.shape[1] == 25360
    for b in range(6):
        rc, a = o.noise_atap(x[b])
        seg = o.vad(x[b], a)
        assert tuple(seg[:2]) == (3200, 3200 + 80 * (T - 1) + 160)
        n, _ = o.mfcc(x[b], seg[0], seg[1], a)
        assert n == T


# ----------------------------------------------------------------------------- tier (ii) == tier (i), live
@needs_reference
def test_tier2_equals_tier1_on_fresh_inputs(oracle):
    r = ol.RefLib()
    rng = np.random.default_rng(77)
    bank = synth.word_bank(7, seed=9)
    B = 24
    frames = rng.integers(20, 120, B)
    pcm = np.zeros((B, 16000), dtype=np.uint16)
    for b in range(B):
        pcm[b] = synth.as_u16_numpy(synth.make_utterances([b % 7], [frames[b]], seed=1000 + b, bank=bank, S=16000,
                                                          gain=float(rng.choice([0.3, 1.0, 1.0, 4.0])),
                                                          quiet_sigma=float(rng.choice([4.0, 8.0]))))[0]
    feats = []
    for b in range(B):
        a1, s1 = r.vad(pcm[b])
        rc, a2 = oracle.noise_atap(pcm[b])
        s2 = oracle.vad(pcm[b], a2)
        assert a1.astuple() == a2.astuple() and np.array_equal(s1, s2)
        if s1[1] < 0 or s1[0] < 1:
            continue
        n1, m1, f1 = r.mfcc(pcm[b], s1[0], s1[1], a1)
        n2, m2 = oracle.mfcc(pcm[b], s2[0], s2[1], a2)
        assert n1 == n2 and np.array_equal(m1, m2)
        if n1:
            feats.append((n1, m1, f1))
    assert len(feats) >= 16
    pad = np.zeros((2, 12), dtype=np.int16)
    for i in range(len(feats)):
        for j in range(len(feats)):
            d1 = r.dtw(feats[i][2], feats[j][2])
            d2 = oracle.dtw(np.concatenate([feats[i][1], pad]), feats[i][0], np.concatenate([feats[j][1], pad]),
                            feats[j][0])
            assert d1 == d2


@needs_reference
def test_tier2_equals_reference_objects_at_benchmark_shape():
    """BASELINE configs[2]'s shape (256-frame utterances, 100 templates of 192..320 frames) through the REFERENCE'S OWN
    compiled VAD.C / MFCC.C / DTW.C: the objects are built with vv_tim_max = 3210 ms instead of 1200 (the one constant
    that caps a record at 119 frames, MFCC.H:15-16; oracle/Makefile writes a patched temporary copy of that header into
    a temporary build directory outside the repository).  The parametrised restatement (tier ii), the oracle of every 256-frame
    GPU test and of bench.py, has to agree with them bit for bit: thresholds, segments, every MFCC vector, all 100
    scores, the argmin -- the same generator and seeds as bench.py."""
    r = ol.RefLib320()
    T, K, NW = 256, 100, 20
    o = ol.Oracle(max_frames=320)
    bank = synth.word_bank(NW)
    rng = np.random.default_rng(2026)
    tfr = rng.integers(192, 321, K)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(K) % NW, tfr, seed=77, bank=bank, S=synth.buf_len_for(320)))
    stride = 8192
    store = np.full(K * stride, 0xFF, dtype=np.uint8)
    tm = np.zeros((K, 321, 12), np.int16)
    for k in range(K):
        a1, s1 = r.vad(tp[k])
        rc, a2 = o.noise_atap(tp[k])
        s2 = o.vad(tp[k], a2)
        assert a1.astuple() == a2.astuple() and np.array_equal(s1, s2)
        n1, m1, f1 = r.mfcc(tp[k], s1[0], s1[1], a1)
        n2, m2 = o.mfcc(tp[k], s2[0], s2[1], a2)
        assert n1 == n2 == tfr[k] and np.array_equal(m1, m2), k
        store[k * stride:k * stride + r.FTR_BYTES] = f1          # save_sign is set by the caller in the firmware (main.c:131)
        store[k * stride:k * stride + 2].view(np.uint16)[0] = 12345
        tm[k, :n1] = m1
    tpl = o.make_templates(tm, tfr.astype(np.uint32))
    B = 64
    words = rng.integers(0, NW, B)
    pcm = synth.as_u16_numpy(synth.make_utterances(words, [T] * B, seed=1000, bank=bank, S=synth.buf_len_for(T)))
    ores, omf, osc = o.recognize_batch(pcm, tpl, n_threads=4)
    n_err = 0
    for b in range(B):
        st, best, dis, scores, mf, n = r.spch_recg(pcm[b], store, stride=stride)
        assert st == 0 and n == T and ores["status"][b] == 0 and ores["frm_num"][b] == T
        assert np.array_equal(mf, omf[b, :T]), b
        assert np.array_equal(scores, osc[b]), b
        assert best == ores["best_tpl"][b] and dis == ores["min_dis"][b], b
        n_err += int((scores == ol.DIS_ERR).sum())
    assert (ores["best_tpl"] % NW == words).mean() > 0.9          # the workload is a recognition task, not noise


@needs_reference
def test_tier2_dtw_equals_tier1_random_features(oracle):
    r = ol.RefLib()
    rng = np.random.default_rng(123)
    for p in range(1500):
        na, nb = (int(v) for v in rng.integers(1, 120, 2))
        amp = int(rng.choice([100, 1500, 32767]))
        a = rng.integers(-amp, amp + 1, (119, 12)).astype(np.int16)
        b = rng.integers(-amp, amp + 1, (119, 12)).astype(np.int16)
        d1 = r.dtw(ol.RefLib.make_ftr(a, na), ol.RefLib.make_ftr(b, nb))
        d2 = oracle.dtw(np.concatenate([a, np.zeros((1, 12), np.int16)]), na,
                        np.concatenate([b, np.zeros((1, 12), np.int16)]), nb)
        assert d1 == d2, (p, na, nb)


@needs_reference
def test_reference_adc_dump_probe_values(oracle):
    """SURVEY.md section 8c probe: the reference's own 12-bit capture 'STM32 123.txt' (read in place)."""
    txt = _ref_text("Matlab/语音样本/STM32 123.txt".encode("utf-8").decode("utf-8"))
    vals = np.array([int(x) for x in re.findall(r"\d+", txt)], dtype=np.uint16)
    assert len(vals) == 16000
    rc, a = oracle.noise_atap(vals)
    assert a.astuple() == (2213, 172, 2, 9524)
    seg = oracle.vad(vals, a)
    assert list(seg) == [3920, 6880, 8640, 11440, 13360, -1]
    n0, m0 = oracle.mfcc(vals, seg[0], seg[1], a)
    n1, m1 = oracle.mfcc(vals, seg[2], seg[3], a)
    assert (n0, n1) == (36, 34)
    assert list(m0[0]) == [353, 719, 516, 439, -174, -29, -18, -48, 29, -113, -23, -5]
    pad = np.zeros((1, 12), np.int16)
    assert oracle.dtw(np.concatenate([m0, pad]), n0, np.concatenate([m1, pad]), n1) == 3874
    r = ol.RefLib()
    a1, s1 = r.vad(vals)
    assert a1.astuple() == a.astuple() and np.array_equal(s1, seg)


# ----------------------------------------------------------------------------- SURVEY 8(f) rows
def test_recognize_segments_matches_golden(oracle, golden):
    """every VAD segment matched like segment 0 (extension of main.c:268); fixture from the reference objects"""
    tm, tf, tv = store_to_templates(golden["store"])
    tpl = oracle.make_templates(tm, tf, tv)
    mw = golden["multi_pcm"]
    for i in range(mw.shape[0]):
        res, sc = oracle.recognize_segments(mw[i], tpl)
        assert np.array_equal(res["status"], golden["multi_status"][i])
        assert np.array_equal(res["frm_num"], golden["multi_frm"][i])
        assert np.array_equal(res["min_dis"], golden["multi_dis"][i])
        assert np.array_equal(res["best_tpl"], golden["multi_best"][i])
        assert np.array_equal(sc, golden["multi_scores"][i])
    assert (golden["multi_status"] == 0).sum() >= 12 and (golden["multi_status"] == 1).sum() >= 2


def test_real_speech_matches_golden(oracle):
    """REAL speech (capture buffers cut from the reference's own recordings, tests/golden/make_real_golden.py) through
    the restatement vs what the reference's compiled objects made of it: noise_atap, all VAD segments, MFCC of every
    segment, all DTW scores against a store trained from real captures, argmin"""
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "real_speech.npz"))
    tm, tf, tv = store_to_templates(g["store"])
    tpl = oracle.make_templates(tm, tf, tv)
    nseg = 0
    for i in range(g["pcm"].shape[0]):
        rc, a = oracle.noise_atap(g["pcm"][i])
        assert a.astuple() == tuple(int(v) for v in g["atap"][i])
        seg = oracle.vad(g["pcm"][i], a)
        assert np.array_equal(seg, g["seg"][i])
        res, sc = oracle.recognize_segments(g["pcm"][i], tpl)
        assert np.array_equal(res["status"], g["status"][i]) and np.array_equal(res["frm_num"], g["frm"][i])
        assert np.array_equal(res["min_dis"], g["dis"][i]) and np.array_equal(res["best_tpl"], g["best"][i])
        assert np.array_equal(sc, g["scores"][i])
        for s_ in range(3):
            if g["status"][i, s_] == 0:
                n, m = oracle.mfcc(g["pcm"][i], int(seg[2 * s_]), int(seg[2 * s_ + 1]), a)
                assert n == g["frm"][i, s_] and np.array_equal(m, g["mfcc"][i, s_, :n])
                nseg += 1
    assert nseg >= 20
    assert (g["dis"][-2:, 0] == 0).all()  # the two template captures match themselves


def expected_store_image(mfcc_rows, frames, slots, n_slots=80, stride=4096, status=None):
    """what save_ftr_mdl leaves in flash (Flash.C:17-67): erased 0xFF slot, then save_mask | frm_num | rows"""
    store = np.full(n_slots * stride, 0xFF, dtype=np.uint8)
    for i, sl in enumerate(slots):
        if status is not None and status[i] != 0:
            continue
        n = int(frames[i])
        img = np.full(stride, 0xFF, dtype=np.uint8)
        img[:4].view(np.uint16)[:] = (12345, n)
        img[4:4 + n * 24] = np.ascontiguousarray(mfcc_rows[i][:n], dtype=np.int16).view(np.uint8).reshape(-1)
        store[sl * stride:(sl + 1) * stride] = img
    return store


def test_store_image_round_trip(oracle, golden):
    """flash-layout image built from MFCC rows parses back to the dense template layout"""
    pcm = golden["pcm"][:6]
    rows, frames = [], []
    for b in range(6):
        n = int(golden["frm_num"][b])
        rows.append(golden["mfcc"][b, :n])
        frames.append(n)
    slots = [0, 5, 17, 42, 60, 79]
    img = expected_store_image(rows, frames, slots)
    tm, tf, tv = store_to_templates(img)
    assert tv.sum() == 6 and [int(tf[s]) for s in slots] == frames
    for s_, r in zip(slots, rows):
        assert np.array_equal(tm[s_, :len(r)], r)
    assert (tm[1, :119] == -1).all() and tv[1] == 0  # erased slot: 0xFFFF halves, not valid


def test_oracle_is_clean_under_asan_ubsan():
    """the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer.  oracle/selftest.c
    drives every entry point (both front ends, the sentinel edge cases, full-scale records, the threaded batch call);
    any out-of-bounds access or undefined operation aborts it."""
    import subprocess
    odir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    b = subprocess.run(["make", "-C", odir, "asan"], capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "sanitize" in b.stderr.lower()):
        pytest.skip("toolchain without sanitizer runtimes: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([os.path.join(odir, "_ref", "oracle_selftest_asan")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "oracle selftest ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_delta_mfcc_definition(oracle, golden):
    """EXTENSION, no reference counterpart: the oracle's delta cepstra against an independent numpy statement of the
    definition (two-frame regression, clamped rows, truncation toward zero) on real MFCC records and full-scale ones."""
    rng = np.random.default_rng(5)
    recs = [(int(golden["frm_num"][b]), golden["mfcc"][b]) for b in range(6) if golden["frm_num"][b] > 0]
    recs += [(n, rng.integers(-32768, 32768, (n, 12)).astype(np.int16)) for n in (1, 2, 3, 4, 5, 40)]
    for n, m in recs:
        m = np.ascontiguousarray(m[:n]).astype(np.int64)
        idx = np.arange(n)
        c = lambda k: m[np.clip(idx + k, 0, n - 1)]
        num = (c(1) - c(-1)) + 2 * (c(2) - c(-2))
        want = (np.sign(num) * (np.abs(num) // 10)).astype(np.int16)
        assert np.array_equal(oracle.delta_mfcc(m.astype(np.int16), n), want), n


def test_extension_fft512_is_a_dft_within_its_truncation_bound(oracle):
    """EXTENSION (no reference counterpart): the 512-point transform the extension front end is defined by
    (oracle/q15_fft.c: two ST-style 256-point radix-4 transforms + one truncating radix-2 pass) is the DFT / 512 of its
    input up to the truncation of its five stages.  Bound: an output component of a radix-4 pass is a sum of three
    floored terms (A>>2, X>>16 or X>>15 twice, .s:105-129), i.e. off by less than 3 LSB, sqrt(2)*3 as a complex number;
    a later pass averages four such inputs with unit-modulus coefficients, so an earlier error reaches the output with
    at most its own size: 4 passes * 4.25 = 17 LSB worst case, + 2 for the two floors of the radix-2 pass, + the Q14
    coefficient rounding, which scales with the signal.  (The errors are one-sided floors of independent data: the
    typical error is a few LSB, asserted as an RMS.)  Checked on random data at several levels, on real zero-padded
    frames (the shape get_mfcc feeds), on impulses and on pure tones."""
    import ctypes as C
    L = oracle.L
    rng = np.random.default_rng(31)

    def run(re, im):
        w = re.astype(np.int16).view(np.uint16).astype(np.uint32) | (im.astype(np.int16).view(np.uint16).astype(np.uint32) << 16)
        out = np.zeros(512, dtype=np.uint32)
        L.sr_oracle_q15_fft512(out.ctypes.data_as(C.c_void_p), np.ascontiguousarray(w).ctypes.data_as(C.c_void_p))
        got = (out & 0xFFFF).astype(np.uint16).view(np.int16).astype(np.float64) + \
            1j * (out >> 16).astype(np.uint16).view(np.int16).astype(np.float64)
        want = np.fft.fft(re.astype(np.float64) + 1j * im.astype(np.float64)) / 512.0
        return got, want

    worst = 0.0
    cases = []
    for amp in (40, 900, 12000, 32767):
        cases.append((rng.integers(-amp, amp + 1, 512), rng.integers(-amp, amp + 1, 512)))
        re = np.zeros(512, np.int64)
        re[:320] = rng.integers(-amp, amp + 1, 320)                      # a real, zero-padded frame
        cases.append((re, np.zeros(512, np.int64)))
    for pos in (0, 1, 255, 511):
        re = np.zeros(512, np.int64)
        re[pos] = 32767
        cases.append((re, np.zeros(512, np.int64)))
    for kbin in (1, 37, 128, 255):
        t = np.arange(512)
        cases.append((np.round(20000 * np.cos(2 * np.pi * kbin * t / 512)).astype(np.int64),
                      np.round(20000 * np.sin(2 * np.pi * kbin * t / 512)).astype(np.int64)))
    for re, im in cases:
        got, want = run(np.asarray(re), np.asarray(im))
        scale = max(np.abs(want).max(), 1.0)
        err = np.abs(got - want)
        assert err.max() <= 19.0 + 6e-4 * scale, (err.max(), scale)
        assert np.sqrt((err ** 2).mean()) <= 5.0 + 3e-4 * scale, (np.sqrt((err ** 2).mean()), scale)
        worst = max(worst, err.max() - 6e-4 * scale)
    assert worst > 0.3                                                    # the bound is not vacuous: truncation is visible
