"""CPU-side checks of the drop-in boundary (no compute calls, no GPU needed):
the in-tree C-ABI library loads and exports every function include/sr_engine.h declares, the struct
layouts match the header, and without a GPU the product refuses to run (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from stm32_speech_recognition_amd import SrError, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sr_engine.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)          # comments
    src = re.sub(r"#pragma[^\n]*", " ", src)
    src = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)  # struct bodies
    names = re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_declares_the_reference_entry_points():
    names = declared_functions()
    for n in ("noise_atap", "VAD", "get_mfcc", "fft", "cr4_fft_1024_stm32", "get_dis", "dtw_limit", "dtw", "spch_recg",
              "GetMfcc", "MFCC_Comp", "sr_create", "sr_set_templates", "sr_recognize_batch", "sr_recognize_batch_dev",
              "sr_mfcc_batch", "sr_dtw_batch", "sr_train_store", "sr_recognize_segments_batch"):
        assert n in names, n
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    L = engine.load_library()
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, missing


HOOKS = ("dtw_u", "dtw_tie_g", "dtw_kc", "mfcc_grid", "perturb_log_thr", "log_thr_from_host", "multi_allow_dup", "dtw_debug",
         "cells_literal", "mag_cheap_off")


def test_product_library_has_no_development_hooks():
    """The shipped libsr_engine.so is built WITHOUT -DSR_TESTING: sr_dev_hook refuses every name, the hook table is not
    in the binary, and the hook-reading branches are compile-time constants.  The suite's hook tests and the tuning sweeps
    use libsr_engine_testing.so (same sources, -DSR_TESTING), which exports the same surface."""
    L = engine.load_library()
    assert os.path.basename(engine.LIB_PATH) == "libsr_engine.so" and L.sr_testing_build() == 0
    for name in HOOKS:
        assert L.sr_dev_hook(name.encode(), C.c_int64(1)) == 3, name          # SR_ERR_BAD_ARG
        assert b"not compiled into the product library" in L.sr_last_error()
    blob = open(engine.LIB_PATH, "rb").read()
    for name in ("cells_literal", "multi_allow_dup", "perturb_log_thr", "log_thr_from_host"):
        assert name.encode() + b"\0" not in blob, name                         # not even the name table
    T = engine.load_library(testing=True)
    assert T.sr_testing_build() == 1
    missing = [n for n in declared_functions() if not hasattr(T, n)]
    assert not missing, missing
    for name in HOOKS:
        assert T.sr_dev_hook(name.encode(), C.c_int64(0)) == 0, name
    assert T.sr_dev_hook(b"no_such_hook", C.c_int64(0)) == 3


def test_reference_header_caller_is_linked_with_the_product_library():
    """tests/ref_caller/ref_caller.c -- the reference's own VAD.H / MFCC.H / DTW.H / ADC.H, the main.c:258-295 sequence,
    -lsr_engine -- is (re)built here whenever the reference tree is present (oracle/Makefile); the GPU suite runs it"""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_caller")
    if os.path.isdir("/root/reference/Src/Speech_Recog"):
        engine.load_library()
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_caller"], stdout=subprocess.DEVNULL)
    if not os.path.exists(exe):
        pytest.skip("no reference tree and no prebuilt oracle/_ref/ref_caller on this box")
    src = open(os.path.join(ROOT, "tests", "ref_caller", "ref_caller.c")).read()
    assert "sr_engine.h\"" not in src.replace("include/sr_engine.h", "") and '#include "VAD.H"' in src and '#include "MFCC.H"' in src
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libsr_engine.so" in needed and "libsr_ref" not in needed and "liboracle" not in needed
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for f in ("noise_atap", "VAD", "get_mfcc", "dtw"):
        assert f" U {f}" in syms, f


def test_struct_layouts_match_header():
    assert engine.RESULT_DTYPE.itemsize == 16 and engine.VAD_DTYPE.itemsize == 48
    assert C.sizeof(engine.Config) == 40
    from stm32_speech_recognition_amd import compat
    assert C.sizeof(compat.v_ftr_tag) == 2860 and C.sizeof(compat.atap_tag) == 12


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(SrError, match="no HIP device|ROCm-capable|gfx950"):
        engine.Engine()


def build_c_demo(out_path, src="spch_recg_demo.c"):
    """examples/spch_recg_demo.c = the reference's main.c:258-283 call pattern, plain C, linked against the library;
    examples/multi_gpu_demo.c = the multi-GPU surface from plain C"""
    import subprocess
    lib_dir = os.path.join(ROOT, "stm32_speech_recognition_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", src), "-L" + lib_dir, "-lsr_engine",
                           "-Wl,-rpath," + lib_dir, "-o", out_path])


def test_c_demo_compiles_and_links_as_plain_c(tmp_path):
    engine.load_library()
    build_c_demo(str(tmp_path / "spch_recg_demo"))
    build_c_demo(str(tmp_path / "multi_gpu_demo"), "multi_gpu_demo.c")


def test_dtw_launch_geometry_respects_the_lds_budget():
    """Host-only sr_dtw_geometry: the staged DTW kernel's workgroup shape for any store size / frame cap.  gfx950 hands out
    LDS in granules of 1280 bytes; in round 3 a workgroup of 53 780 bytes (43 granules instead of 42) silently cost the third
    workgroup per CU.  Invariants: U * Kc <= 1024 lanes, the workgroups the scorer counted on really fit 160 KiB at the
    granule, the table is a power of two >= 4096; and the benchmark shapes keep their measured optima."""
    import ctypes as C
    from stm32_speech_recognition_amd.engine import load_library
    L = load_library()
    out = (C.c_uint32 * 5)()

    def geo(K, R):
        assert L.sr_dtw_geometry(C.c_uint32(K), C.c_uint32(R), out) == 0
        return tuple(out)

    for R in (2, 16, 48, 119, 257, 320, 900, 2000, 5000):
        for K in (1, 3, 10, 80, 100, 130, 500, 513, 700, 1024, 1500, 2050, 5000):
            U, kc, g, lds, wgs = geo(K, R)
            assert U >= 1 and 1 <= kc <= K and U * kc <= 1024, (K, R, U, kc)
            assert g >= 4096 and g & (g - 1) == 0 and g <= 32768
            gran = (lds + 1279) // 1280
            assert wgs >= 1 and wgs * gran * 1280 <= 160 * 1024, (K, R, lds, wgs)
    assert geo(100, 6000)[0] == 0                                   # no room for one utterance: generic kernel
    assert geo(100, 320)[:3] == (5, 100, 8192) and geo(100, 320)[4] == 3       # BASELINE configs[2]: three workgroups per CU
    U, kc, g, lds, wgs = geo(500, 320)                                          # configs[4]: several lanes per template row
    assert U >= 6 and kc <= 170 and g >= 16384 and wgs == 2
