"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle and the
committed golden fixtures (which come from the reference's own objects).  Bit-exact everywhere:
the whole path is integer except sqrtf / log, which are reproduced exactly.
"""
import os
import sys

import numpy as np
import pytest
import torch

import oracle_lib as ol
from stm32_speech_recognition_amd import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def eng119():
    from stm32_speech_recognition_amd import Engine
    e = Engine(max_frames=119, device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def oracle():
    return ol.Oracle(max_frames=119)


def _dtw_all_modes(eng, im, inf):
    """sr_dtw_batch with the batch kernels (small-launch mode 1), with one workgroup per pair wherever the band fits
    (mode 2, k_dtw_cells), with four lanes per pair (mode 3, k_dtw_quad), in the automatic mode and with k_dtw_cells' literal
    fallback forced: scores and results must be the same bytes; returns the first"""
    from stm32_speech_recognition_amd.engine import dev_hook
    eng.set_small_launch(1)
    sc, res = eng.dtw(im, inf)
    for mode in (2, 3, 0):
        eng.set_small_launch(mode)
        sc2, res2 = eng.dtw(im, inf)
        assert np.array_equal(sc, sc2), mode
        assert res.tobytes() == res2.tobytes(), mode
    # k_dtw_cells' fallback for walks that leave dtw_limit's band (a step with all three candidates outside): the literal
    # walk on the staged rows, forced for every pair by the development hook -- which exists only in the -DSR_TESTING build
    # of the library, so an engine with the same configuration and store is opened there
    et = eng.clone(testing=True)
    dev_hook("cells_literal", 1)
    try:
        et.set_small_launch(2)
        sc3, res3 = et.dtw(im, inf)
    finally:
        dev_hook("cells_literal", 0)
        et.close()
        eng.set_small_launch(0)
    assert np.array_equal(sc, sc3) and res.tobytes() == res3.tobytes()
    return sc, res


def test_library_loaded_is_in_tree():
    from stm32_speech_recognition_amd import engine
    engine.load_library()
    assert os.path.exists(engine.LIB_PATH)
    assert "libsr_engine.so" in open("/proc/self/maps").read()


# ----------------------------------------------------------------------------- golden fixtures
def test_fft_q15_matches_golden(eng119, golden):
    """generic complex FFT (cr4_fft_1024_stm32 symbol) incl. full-scale inputs"""
    got = eng119.fft_q15(golden["fft_in"])
    assert np.array_equal(got, golden["fft_out"])


@pytest.mark.parametrize("mode", [1, 0])
def test_vad_matches_golden(eng119, golden, mode):
    """mode 1: one wave per capture (k_vad); mode 0: the fixture has fewer captures than CUs -> four waves per capture (k_vad_wide)"""
    eng119.set_small_launch(mode)
    vd = eng119.vad(golden["pcm"])
    eng119.set_small_launch(0)
    assert np.array_equal(np.stack([vd["mid_val"], vd["n_thl"], vd["z_thl"], vd["s_thl"]], 1), golden["atap"])
    assert np.array_equal(vd["seg"], golden["seg"])
    assert np.array_equal(vd["frm_num"], golden["frm_num"])


def test_recognize_matches_golden(eng119, golden):
    eng119.set_templates_store(golden["store"])
    out = eng119.recognize(golden["pcm"])
    n = golden["frm_num"]
    for b in range(len(n)):
        assert np.array_equal(out["mfcc"][b, :n[b]], golden["mfcc"][b, :n[b]]), b
        assert not out["mfcc"][b, n[b]:].any()
    assert np.array_equal(out["scores"], golden["recg_scores"])
    assert np.array_equal(out["results"]["best_tpl"], golden["recg_best"])
    assert np.array_equal(out["results"]["min_dis"], golden["recg_dis"])
    assert np.array_equal(out["results"]["status"], golden["recg_status"])


def test_direct_mfcc_edge_cases_match_golden(eng119, golden):
    """log(0) frames, s16 wrap after windowing, u32 wrap in the filterbank, full-range u16 codes"""
    dp, dmid, dm = golden["direct_pcm"], golden["direct_mid"], golden["direct_mfcc"]
    D, nfd = dm.shape[0], dm.shape[1]
    for mode in (0, 1):  # through the pinned staging area / with blocking copies
        eng119.set_small_launch(mode)
        n, m = eng119.mfcc(dp, np.ones(D, np.int32), np.full(D, 1 + 160 + 80 * (nfd - 1), np.int32), dmid)
        assert (n == nfd).all()
        assert np.array_equal(m[:, :nfd], dm)
    eng119.set_small_launch(0)


def test_dtw_matches_golden(golden):
    """400 random pairs incl. every length gate and full-scale coefficients, one template per engine call"""
    from stm32_speech_recognition_amd import Engine
    ln, da, db, dd = golden["dtw_len"], golden["dtw_a"], golden["dtw_b"], golden["dtw_dis"]
    e = Engine(max_frames=119, device=0)
    # all models as one store, all inputs as one batch: score[p, p] is the pair's distance
    P = len(dd)
    e.set_templates_dense(np.concatenate([db, np.zeros((P, 1, 12), np.int16)], 1), ln[:, 1])
    sc, _ = _dtw_all_modes(e, da, ln[:, 0])
    assert np.array_equal(np.diagonal(sc), dd)
    e.close()


def test_get_mdl_matches_golden(golden, oracle):
    """template averaging (get_mdl, DTW.C:217-296) against the reference's own get_mdl: merged frames, count, distance"""
    from stm32_speech_recognition_amd import Engine, compat
    ln, da, db = golden["dtw_len"], golden["dtw_a"], golden["dtw_b"]
    md, mn, mm = golden["mdl_dis"], golden["mdl_frames"], golden["mdl_rows"]
    P = len(md)
    z = np.zeros((P, 1, 12), np.int16)
    e = Engine(max_frames=119, device=0)
    mdl, frames, dis = e.get_mdl(np.concatenate([da[:P], z], 1), ln[:P, 0], np.concatenate([db[:P], z], 1), ln[:P, 1], 238)
    assert np.array_equal(dis, md) and np.array_equal(frames, mn) and np.array_equal(mdl, mm)
    # clipped output: the count still reports the full merged length
    mdl50, frames50, dis50 = e.get_mdl(np.concatenate([da[:P], z], 1), ln[:P, 0], np.concatenate([db[:P], z], 1), ln[:P, 1], 50)
    assert np.array_equal(frames50, mn) and np.array_equal(dis50, md) and np.array_equal(mdl50, mm[:, :50])
    e.close()
    # the reference symbols themselves (v_ftr_tag records), where the merged template fits the record
    done = 0
    for p in range(P):
        if md[p] == ol.DIS_ERR or mn[p] > 119 or max(ln[p]) == 119:
            continue
        dis1, out = compat.get_mdl(compat.make_ftr(da[p], int(ln[p, 0])), compat.make_ftr(db[p], int(ln[p, 1])))
        got = np.ctypeslib.as_array(out.mfcc_dat)[:out.frm_num * 12].reshape(-1, 12)
        assert dis1 == md[p] and out.frm_num == mn[p] and np.array_equal(got, mm[p, :mn[p]]), p
        done += 1
        if done == 12:
            break
    assert done == 12
    p = int(np.argmax(md == ol.DIS_ERR))
    assert compat.get_mdl(compat.make_ftr(da[p], int(ln[p, 0])), compat.make_ftr(db[p], int(ln[p, 1])))[0] == ol.DIS_ERR
    a, b = np.array([-32768, 32767, -3, 3, -1, 1, 0, 5, -5, 101, -101, 7], np.int16), \
        np.array([-32768, 32767, 0, 0, 0, 0, 1, 0, 0, 0, 0, -8], np.int16)
    want = np.array([int((int(x) + int(y)) / 2) for x, y in zip(a, b)], np.int16)  # C division truncates toward zero
    assert np.array_equal(compat.get_mean(a, b), want)


def test_get_mdl_long_records_match_oracle():
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(77)
    P, R = 64, 321
    orc = ol.Oracle(max_frames=320)
    n1 = rng.integers(1, 321, P).astype(np.uint32)
    n2 = np.array([rng.integers(max(1, (int(v) + 1) // 2), min(320, 2 * int(v)) + 1) for v in n1], np.uint32)
    n2[::9] = rng.integers(1, 321, len(n2[::9]))
    a = rng.integers(-3000, 3000, (P, R, 12)).astype(np.int16)
    b = (a[:, rng.permutation(R)] + rng.integers(-400, 400, (P, R, 12))).astype(np.int16)
    a[::7] = rng.integers(-32768, 32768, (len(a[::7]), R, 12))
    e = Engine(max_frames=320, device=0)
    mdl, frames, dis = e.get_mdl(a, n1, b, n2, 640)
    for p in range(P):
        d, n, rows = orc.get_mdl(a[p], n1[p], b[p], n2[p], 640)
        assert d == dis[p] and n == frames[p] and np.array_equal(rows, mdl[p, :n]) and not mdl[p, n:].any(), p
    assert (dis != ol.DIS_ERR).sum() > 40
    e.close()


# ----------------------------------------------------------------------------- oracle on fresh inputs
def _oracle_templates(orc, bank, frames, seed, S):
    K = len(frames)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(K) % bank[0].shape[0], frames, seed=seed, bank=bank, S=S))
    tm = np.zeros((K, max(frames) + 1, 12), np.int16)
    for k in range(K):
        rc, a = orc.noise_atap(tp[k])
        seg = orc.vad(tp[k], a)
        n, m = orc.mfcc(tp[k], seg[0], seg[1], a)
        assert n == frames[k]
        tm[k, :n] = m
    return tm, np.array(frames, np.uint32)


@pytest.mark.parametrize("T,B,K", [(256, 1024, 100), (256, 4096, 10), (119, 512, 10)])
def test_full_path_matches_oracle(T, B, K):
    """BASELINE configs[1] in full (4096 x 10, every utterance compared) and a 1024-utterance sample of
    configs[2] (x 100 templates), 256 frames each, plus a reference-sized case: MFCC s16 exact, all K scores u32
    exact, argmin exact (>= 1024 utterances per config)."""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import results_from_torch, vad_from_torch
    rng = np.random.default_rng(T + B)
    maxf = T + 64
    orc = ol.Oracle(max_frames=maxf)
    bank = synth.word_bank(10)
    tfr = [int(v) for v in rng.integers(int(0.75 * T), int(1.25 * T), K)]
    tfr[3] = max(2, T // 2 - 5)  # outside the 1/2..2x gate of DTW.C:133 -> dis_err
    tm, tf = _oracle_templates(orc, bank, tfr, seed=7, S=synth.buf_len_for(max(tfr)))
    valid = np.ones(K, np.uint8)
    valid[5] = 0
    S = synth.buf_len_for(T)
    pcm_t = synth.make_utterances(rng.integers(0, 10, B), [T] * B, seed=99, bank=bank, S=S, device="cuda:0")
    # a few rows with noisy quiet parts (segment length varies), one silent row (VAD fail)
    noisy = synth.make_utterances(rng.integers(0, 10, 8), [T - 20] * 8, seed=5, bank=bank, S=S, quiet_sigma=8.0)
    pcm_t[:8] = noisy.to("cuda:0")
    pcm_t[8] = 2048
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf, valid)
    out = eng.recognize_dev(pcm_t, eng.alloc_outputs(B, "cuda:0"))
    torch.cuda.synchronize()
    res = results_from_torch(out["results"])
    vd = vad_from_torch(out["vad"])
    tpl = orc.make_templates(tm, tf, valid)
    ores, omf, osc = orc.recognize_batch(synth.as_u16_numpy(pcm_t), tpl, n_threads=min(64, os.cpu_count() or 8))
    assert np.array_equal(vd["status"], ores["status"])
    assert np.array_equal(out["mfcc"].cpu().numpy(), omf)
    assert np.array_equal(out["scores"].cpu().numpy().view(np.uint32), osc)
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(res[f], ores[f]), f
    assert res["status"][8] == ol.ST_VAD_FAIL and res["min_dis"][8] == ol.DIS_ERR
    assert (osc[:, 5] == ol.DIS_ERR).all() and (osc[9:, 3] == ol.DIS_ERR).all()
    assert (res["frm_num"][9:] == T).all()
    eng.close()


@pytest.mark.parametrize("gain", [1.0, 2.4, 4.0])
def test_full_path_matches_reference_objects(gain):
    """(gain: the generator's speech amplitude -- 1.0 = bench.py's headline, 2.4 = SURVEY.md 8(d)'s 200-600, 4.0 = most frames in
    the frame kernel's LOUD tier; the three tiers of k_mfcc's magnitude / filterbank stage, DESIGN.md 3.2, are all on the path.)
    Tier (i) at the BENCHMARK shape inside the GPU suite: the HIP path against the reference's OWN VAD.C / MFCC.C / DTW.C
    objects (oracle/_ref/libsr_ref320.so: compiled from the reference tree with the one constant that caps a record at
    119 frames raised to 320, MFCC.H:15-16; oracle/Makefile) -- 256-frame utterances x 100 templates of 192..320 frames,
    one template outside the 1/2..2x gate of DTW.C:133, one erased slot, some captures with ragged segments and one
    silent capture.  Templates come from the same objects (main.c:121-138).  MFCC rows, every score, argmin and min_dis
    must be identical (main.c:249-296 restated over an explicit store in oracle/ref_glue.c)."""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import results_from_torch, vad_from_torch
    if not ol.RefLib320.available():
        pytest.fail("oracle/_ref/libsr_ref320.so is missing: build it where /root/reference exists (make -C oracle); it travels with the snapshot")
    T, B, K = 256, 1024, 100
    rng = np.random.default_rng(2605)
    bank = synth.word_bank(20)
    ref = ol.RefLib320()
    tfr = [int(v) for v in rng.integers(192, 321, K)]
    tfr[3] = 120   # 2 * 120 < 256: in > 2 * mdl -> dis_err (DTW.C:133-137)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(K) % 20, tfr, seed=77, bank=bank, S=synth.buf_len_for(320), gain=gain))
    tm = np.zeros((K, 321, 12), np.int16)
    for k in range(K):
        a, seg = ref.vad(tp[k])
        n, m, _ = ref.mfcc(tp[k], int(seg[0]), int(seg[1]), a)
        assert n == tfr[k]
        tm[k, :n] = m
    tf = np.array(tfr, np.uint32)
    valid = np.ones(K, np.uint8)
    valid[5] = 0   # erased flash slot: save_sign != 12345 (main.c:283)
    S = synth.buf_len_for(T)
    pcm_t = synth.make_utterances(rng.integers(0, 20, B), [T] * B, seed=99, bank=bank, S=S, device="cuda:0", gain=gain)
    noisy = synth.make_utterances(rng.integers(0, 20, 8), [T - 20] * 8, seed=5, bank=bank, S=S, quiet_sigma=8.0, gain=gain)
    pcm_t[:8] = noisy.to("cuda:0")
    pcm_t[8] = 2048
    eng = Engine(max_frames=320, device=0)
    assert eng.mag_cheap_bound() == 70171   # sr_create's sweep of this device confirmed the cheap magnitude form (else: 0, exact roots)
    eng.set_templates_dense(tm, tf, valid)
    out = eng.recognize_dev(pcm_t, eng.alloc_outputs(B, "cuda:0"))
    torch.cuda.synchronize()
    tiers = ol.Oracle(max_frames=320).frame_tiers(synth.as_u16_numpy(pcm_t[16:48]))
    if gain == 1.0:
        assert tiers["quiet"] > 0.9
    elif gain == 2.4:
        assert min(tiers["quiet"], tiers["mid"], tiers["loud"]) > 0.1, tiers   # every tier carries a real share of the frames
    else:
        assert tiers["loud"] > 0.5, tiers
    res = results_from_torch(out["results"])
    gmf = out["mfcc"].cpu().numpy()
    gsc = out["scores"].cpu().numpy().view(np.uint32)
    host = synth.as_u16_numpy(pcm_t)
    pool = ol.Ref320Pool(min(64, os.cpu_count() or 8))
    try:
        rr = pool.recognize(host, ol.ref320_store(tm, tf, valid), K, want_mfcc=True)
    finally:
        pool.close()
    assert "libsr_ref320" in open("/proc/self/maps").read()   # the reference's objects are what this process compared with
    ok = rr["status"] == 0
    assert ok.sum() >= B - 1 and rr["status"][8] == 1 and res["status"][8] == ol.ST_VAD_FAIL and res["min_dis"][8] == ol.DIS_ERR
    assert np.array_equal(res["status"] == 0, ok)
    assert np.array_equal(res["frm_num"][ok], rr["frm_num"][ok]) and (res["frm_num"][9:] == T).all()
    assert len(set(res["frm_num"][:8])) > 1 or res["frm_num"][0] != T   # the ragged rows really are ragged
    assert np.array_equal(gmf[ok], rr["mfcc"][ok])
    assert np.array_equal(gsc[ok], rr["scores"][ok])
    assert np.array_equal(res["best_tpl"][ok], rr["best"][ok]) and np.array_equal(res["min_dis"][ok], rr["dis"][ok])
    assert (gsc[ok][:, 5] == ol.DIS_ERR).all() and (gsc[9:, 3] == ol.DIS_ERR).all()
    eng.close()


def test_chunked_multi_stream_pipeline_matches_oracle():
    """sr_recognize_batch_dev cuts large batches into chunks on internal streams; force that path at a small batch
    (8 chunks of 96 utterances over 4 streams, ragged last chunk) and compare everything with the oracle"""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import results_from_torch
    T, B, K = 119, 8 * 96 + 5, 10
    bank = synth.word_bank(8)
    orc = ol.Oracle(max_frames=T)
    S = synth.buf_len_for(T)
    tm, tf = _oracle_templates(orc, bank, [int(v) for v in np.random.default_rng(3).integers(60, T + 1, K)], seed=9, S=S)
    rng = np.random.default_rng(12)
    frames = rng.integers(40, T + 1, B)
    pcm_t = synth.make_utterances(rng.integers(0, 8, B), frames, seed=4, bank=bank, S=S)
    pcm = synth.as_u16_numpy(pcm_t)
    pcm[5] = 2048  # a capture with no speech at all -> VAD fail in the middle of a chunk
    eng = Engine(max_frames=T, device=0)
    eng.set_pipeline(streams=3, min_chunk=96, max_chunks=12)
    eng.set_templates_dense(tm, tf)
    dev = torch.device("cuda", 0)
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    d_pcm = torch.from_numpy(pcm.view(np.int16)).to(dev)
    torch.cuda.synchronize()  # the upload above ran on the default stream
    eng.set_profiling(True)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # a non-default caller stream: fork / join must order against it
        eng.recognize_dev(d_pcm, out)
        res_t = out["results"].clone()
    side.synchronize()
    st = eng.stage_ms()
    eng.set_profiling(False)
    assert st["launches_per_call"] == 8 and st["total"] > 0  # 773 // 96 = 8 chunks (< max_chunks)
    tpl = orc.make_templates(tm, tf)
    ores, omf, osc = orc.recognize_batch(pcm, tpl, n_threads=8)
    res = results_from_torch(res_t)
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(res[f], ores[f]), f
    assert np.array_equal(out["scores"].cpu().numpy().view(np.uint32), osc)
    assert np.array_equal(out["mfcc"].cpu().numpy(), omf)
    assert (ores["status"] != 0).sum() >= 1
    eng.close()


def test_full_size_batch_properties():
    """BASELINE configs[2] at its full size (65 536 utterances x 100 templates x 256 frames), checked through
    size-independent properties: (a) chunked 3-stream execution == one chunk on one stream, bit for bit; (b) reversing
    the batch order reverses the outputs and changes nothing else; (c) template captures mixed into the batch match
    themselves with distance 0; (d) a strided sample of 384 utterances (first / last of every chunk included)
    equals the oracle in MFCC, all 100 scores and argmin."""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import results_from_torch, vad_from_torch
    dev = torch.device("cuda", 0)
    T, B, K, NW = 256, 65536, 100, 25
    rng = np.random.default_rng(2026)
    bank = synth.word_bank(NW)
    eng = Engine(max_frames=320, device=0)
    tfr = rng.integers(192, 321, K)
    S = synth.buf_len_for(320)
    tp = synth.make_utterances(np.arange(K) % NW, tfr, seed=77, bank=bank, S=S, device=dev)
    tvad, tmf = eng.features_dev(tp)
    torch.cuda.synchronize()
    assert np.array_equal(vad_from_torch(tvad)["frm_num"], tfr)
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    pcm = synth.make_utterances(rng.integers(0, NW, B), [T] * B, seed=1000, bank=bank, S=S, device=dev)
    pcm[1000:1000 + K] = tp  # (c) the template captures themselves, same buffer length
    out = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    eng.recognize_dev(pcm, out)
    torch.cuda.synchronize()
    res = results_from_torch(out["results"])
    # (a)
    eng.set_pipeline(streams=1)
    out1 = eng.alloc_outputs(B, dev, mfcc=True, vad=True)
    eng.recognize_dev(pcm, out1)
    torch.cuda.synchronize()
    eng.set_pipeline()
    for k in ("results", "scores", "mfcc", "vad"):
        assert torch.equal(out[k], out1[k]), k
    # (b)
    rev = torch.flip(pcm, dims=[0]).contiguous()
    eng.recognize_dev(rev, out1)
    torch.cuda.synchronize()
    assert torch.equal(torch.flip(out1["results"], dims=[0]), out["results"])
    assert torch.equal(torch.flip(out1["scores"], dims=[0]), out["scores"])
    del rev, out1
    # (c)
    own = res[1000:1000 + K]
    assert (own["status"] == 0).all() and (own["min_dis"] == 0).all() and np.array_equal(own["frm_num"], tfr)
    sc_own = out["scores"][1000:1000 + K].cpu().numpy().view(np.uint32)
    assert (np.diagonal(sc_own) == 0).all()
    assert (res["status"] == 0).all()
    # (d)
    per = (B + 11) // 12
    idx = sorted(set(list(range(0, B, 173))[:320] + [c * per + d for c in range(12) for d in (0, 1, per - 1) if c * per + d < B]
                     + [1000, 1001, B - 1]))
    sel = torch.tensor(idx, device=dev)
    host = synth.as_u16_numpy(pcm[sel].cpu())
    orc = ol.Oracle(max_frames=320)
    tpl = orc.make_templates(tm, tfr.astype(np.uint32))
    ores, omf, osc = orc.recognize_batch(host, tpl, n_threads=16)
    assert np.array_equal(out["scores"][sel].cpu().numpy().view(np.uint32), osc)
    assert np.array_equal(out["mfcc"][sel].cpu().numpy(), omf)
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(res[f][idx], ores[f]), f
    eng.close()


@pytest.mark.parametrize("seed,maxf,K", [(1, 257, 64), (2, 119, 101), (3, 37, 3), (4, 150, 1), (5, 64, 130), (6, 16, 17),
                                          (7, 900, 20), (8, 2000, 12), (9, 5000, 5), (10, 6000, 4)])
def test_random_shapes_match_oracle(seed, maxf, K):
    """seeded random engine shapes: frame cap, template count (incl. 1 and counts that leave DTW workgroups ragged),
    batch size, utterance lengths that exceed the cap (MFCC fail), silent captures, erased slots, forced chunking.
    The large frame caps walk through the LDS budget of the staged DTW kernel: 900 / 2000 rows per utterance leave room for
    5 / 2 utterances per workgroup and different tie-table sizes, 5000 for one (a 144 KB workgroup), 6000 for none
    (generic kernel)."""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import results_from_torch, vad_from_torch
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 260))
    tmax = max(2, min(maxf, 200))
    S = synth.buf_len_for(int(tmax * 1.3) + 2)
    bank = synth.word_bank(7)
    orc = ol.Oracle(max_frames=maxf)
    tfr = [int(v) for v in rng.integers(max(1, tmax // 2), tmax + 1, K)]
    tm, tf = _oracle_templates(orc, bank, tfr, seed=seed, S=S)
    tm = tm[:, :maxf + 1]
    if tm.shape[1] < maxf + 1:
        tm = np.concatenate([tm, np.zeros((K, maxf + 1 - tm.shape[1], 12), np.int16)], 1)
    valid = (rng.random(K) > 0.1).astype(np.uint8)
    frames = rng.integers(1, int(tmax * 1.3) + 1, B)          # some longer than the cap -> status MFCC_FAIL
    pcm_t = synth.make_utterances(rng.integers(0, 7, B), frames, seed=50 + seed, bank=bank, S=S)
    pcm = synth.as_u16_numpy(pcm_t)
    for b in rng.integers(0, B, max(1, B // 20)):
        pcm[b] = 2048 + (b % 3)                                # silent captures -> VAD fail
    eng = Engine(max_frames=maxf, device=0)
    eng.set_pipeline(streams=3, min_chunk=int(rng.choice([7, 64, 4096])), max_chunks=12)
    eng.set_templates_dense(tm, tf, valid)
    d_pcm = torch.from_numpy(pcm.view(np.int16)).to("cuda:0")
    out = eng.recognize_dev(d_pcm, eng.alloc_outputs(B, "cuda:0"))
    torch.cuda.synchronize()
    res = results_from_torch(out["results"])
    vd = vad_from_torch(out["vad"])
    tpl = orc.make_templates(tm, tf, valid)
    ores, omf, osc = orc.recognize_batch(pcm, tpl, n_threads=8)
    assert np.array_equal(vd["status"], ores["status"]), (maxf, K, B)
    assert np.array_equal(out["mfcc"].cpu().numpy(), omf), (maxf, K, B)
    assert np.array_equal(out["scores"].cpu().numpy().view(np.uint32), osc), (maxf, K, B)
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(res[f], ores[f]), (f, maxf, K, B)
    # the same call with one workgroup per pair (k_dtw_cells) wherever the in x mdl rectangle fits, with four lanes per pair
    # (k_dtw_quad), and with both switched off
    for mode in (2, 3, 1):
        eng.set_small_launch(mode)
        out2 = eng.recognize_dev(d_pcm, eng.alloc_outputs(B, "cuda:0"))
        torch.cuda.synchronize()
        assert np.array_equal(out2["scores"].cpu().numpy().view(np.uint32), osc), (mode, maxf, K, B)
        assert torch.equal(out2["results"], out["results"]), (mode, maxf, K, B)
    eng.close()


def test_bench_two_ranks_end_to_end(tmp_path):
    """bench.py's N = 2 code path end to end on this 1-GPU box: two ranks launched by torch.distributed.run share
    device 0 and use gloo for the collectives (test hooks SR_BENCH_BACKEND / SR_BENCH_DEVICE); exercises the shards,
    the double-buffered score exchange, barriers, max-over-ranks timing and the JSON contract"""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SR_BENCH_BACKEND="gloo", SR_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4096", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[:3000] + "\n...\n" + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["batch_per_gpu"] == 4096 and "all-gather" in j["config"]["parallelism"]
    assert abs(j["value"] - 2 * 4096 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
    assert j["top1_word_accuracy"] == 1.0


def test_vad_stress_matches_oracle(eng119, oracle):
    """random band-crossing activity: tight thresholds, DC steps, bursts -> exercises the block-summary
    reconstruction of last_sig (VAD.C:99,131-157) against the sample-by-sample oracle"""
    rng = np.random.default_rng(2024)
    B, S = 96, 16000
    pcm = np.zeros((B, S), np.uint16)
    for b in range(B):
        sig = rng.normal(0, rng.choice([2, 8, 30]), S)
        nburst = rng.integers(0, 12)
        for _ in range(nburst):
            p = rng.integers(2400, S - 200)
            ln = rng.integers(40, 3000)
            amp = rng.choice([15, 40, 200, 1500])
            f = rng.uniform(50, 3900)
            seg = amp * np.sin(2 * np.pi * f * np.arange(ln) / 8000 + rng.uniform(0, 6.28))
            sig[p:p + ln] += seg[:max(0, min(ln, S - p))]
        if b % 7 == 0:
            sig[rng.integers(2400, S):] += rng.choice([-60, 60])  # DC step: band re-entry from one side only
        pcm[b] = np.clip(2048 + sig, 0, 4095).astype(np.uint16)
    eng119.set_small_launch(1)   # one wave per capture (k_vad), whatever the launch size
    vd = eng119.vad(pcm)
    eng119.set_small_launch(0)   # fewer captures than CUs: four waves per capture (k_vad_wide)
    vdw = eng119.vad(pcm)
    assert vd.tobytes() == vdw.tobytes()
    nseg = 0
    for b in range(B):
        rc, a = oracle.noise_atap(pcm[b])
        seg = oracle.vad(pcm[b], a)
        assert (vd["mid_val"][b], vd["n_thl"][b], vd["z_thl"][b], vd["s_thl"][b]) == a.astuple(), b
        assert np.array_equal(vd["seg"][b], seg), (b, vd["seg"][b], seg)
        nseg += int((seg[1::2] >= 0).sum())
    assert nseg > 40


@pytest.mark.parametrize("tpl_lo,tpl_hi", [(-32768, 32767), (-16383, 16384)])
def test_dtw_stress_matches_oracle(tpl_lo, tpl_hi):
    """random s16 records, including full-scale ones.  The LDS-staged kernel stores templates as -2*coef, which
    needs coefficients in [-16383, 16384]: the second case stays inside that range (both ends present) and runs
    staged against full-scale inputs; the first case has coefficients outside it and must fall to k_dtw."""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(31)
    maxf, K, B = 200, 37, 64
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(1, maxf, K).astype(np.uint32)
    tm = rng.integers(-3000, 3000, (K, maxf + 1, 12)).astype(np.int16)
    tm[::5] = rng.integers(tpl_lo, tpl_hi + 1, (len(tm[::5]), maxf + 1, 12))
    tm[1, :, 0], tm[1, :, 1] = tpl_lo, tpl_hi
    inf = rng.integers(1, maxf + 1, B).astype(np.uint32)
    im = rng.integers(-3000, 3000, (B, maxf, 12)).astype(np.int16)
    im[::4] = rng.integers(-32768, 32767, (len(im[::4]), maxf, 12))
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf)
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, 12), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) for k in range(K)] for b in range(B)],
                    dtype=np.uint32)
    assert np.array_equal(sc, want)
    wb = np.array([int(np.argmin(w)) if w.min() != ol.DIS_ERR else 0 for w in want])
    assert np.array_equal(res["best_tpl"], wb) and np.array_equal(res["min_dis"], want.min(1))
    eng.close()


def test_dtw_ties_and_perfect_squares_match_oracle():
    """records whose frames differ in one or two small coefficients: the squared distances are small perfect squares
    and sums of two squares, so candidates tie constantly and sit exactly on the (g+1)^2 boundaries the staged
    kernel's one-root step compares against (tie order diagonal > up > right, DTW.C:168-184)"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(8)
    maxf, K, B = 120, 50, 40
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(30, maxf, K).astype(np.uint32)
    inf = rng.integers(40, 100, B).astype(np.uint32)
    tm = np.zeros((K, maxf + 1, 12), np.int16)
    im = np.zeros((B, maxf, 12), np.int16)
    tm[:, :, 0] = rng.integers(0, 12, (K, maxf + 1))
    im[:, :, 0] = rng.integers(0, 12, (B, maxf))
    tm[::3, :, 5] = rng.integers(-3, 4, (len(tm[::3]), maxf + 1))
    im[::2, :, 5] = rng.integers(-3, 4, (len(im[::2]), maxf))
    tm[1::7, :, 0] *= 4097          # large perfect squares around and above 2^24
    im[1::5, :, 0] *= 4097
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf)
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, 12), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) for k in range(K)] for b in range(B)],
                    dtype=np.uint32)
    assert np.array_equal(sc, want)
    assert (want != ol.DIS_ERR).sum() > 500
    eng.close()


@pytest.mark.parametrize("scale", [600, 1500, 4000, 9000])
def test_dtw_near_ties_at_large_roots_match_oracle(scale):
    """every frame = one large base vector + a perturbation of a few units in two coefficients: the three squared
    candidates of a step lie within tens of each other at 2^24 .. 2^30, where (float)d rounds and the step T(g) of the
    root function falls short of (g+1)^2 -- the range the staged kernel's tie-threshold table covers (scale 9000 also
    crosses its end, g = 32768, into the literal path)"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(scale)
    maxf, K, B = 90, 40, 48
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(40, maxf, K).astype(np.uint32)
    inf = rng.integers(45, 85, B).astype(np.uint32)
    base_t = rng.integers(-scale, scale, (K, 1, 12))
    base_i = rng.integers(-scale, scale, (B, 1, 12))
    tm = np.repeat(base_t, maxf + 1, 1)
    im = np.repeat(base_i, maxf, 1)
    tm[:, :, 3] += rng.integers(-2, 3, (K, maxf + 1))
    tm[:, :, 7] += rng.integers(-1, 2, (K, maxf + 1))
    im[:, :, 3] += rng.integers(-2, 3, (B, maxf))
    im[:, :, 9] += rng.integers(-1, 2, (B, maxf))
    tm, im = tm.astype(np.int16), im.astype(np.int16)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf)
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, 12), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) for k in range(K)] for b in range(B)],
                    dtype=np.uint32)
    assert np.array_equal(sc, want)
    ok = want != ol.DIS_ERR
    assert ok.sum() > 500 and np.median(want[ok]) > 1.5 * scale
    eng.close()


@pytest.mark.parametrize("seed", range(6))
def test_small_launch_dtw_matches_oracle(seed):
    """launches of a few pairs -- spch_recg's one capture against the store (main.c:276-295), single dtw() calls -- are scored by
    k_dtw_cells, one workgroup per pair: every point's candidates / minimum / move first, then one lane follows the moves.
    Random shapes around the corners of DTW.C:120-192: 1- and 2-frame sequences (the do-while reads the slack row), every
    length gate, erased slots, inputs at the frame cap, ties (small coefficients) and full-scale rows; automatic mode with
    B * K <= 1024, checked against the oracle and against the batch kernels"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(700 + seed)
    maxf = int(rng.choice([2, 3, 17, 60, 119, 150]))
    K = int(rng.choice([1, 2, 9, 80]))
    B = int(rng.integers(1, max(2, 1024 // K // 4)))
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(1, maxf + 1, K).astype(np.uint32)
    inf = rng.integers(1, maxf + 1, B).astype(np.uint32)
    tf[: min(K, 3)] = [1, min(2, maxf), maxf][: min(K, 3)]
    inf[: min(B, 3)] = [maxf, 1, min(2, maxf)][: min(B, 3)]
    amp = int(rng.choice([6, 3000, 32767]))
    tm = rng.integers(-amp, amp + 1, (K, maxf + 1, 12)).astype(np.int16)
    im = rng.integers(-amp, amp + 1, (B, maxf, 12)).astype(np.int16)
    valid = (rng.random(K) > 0.15).astype(np.uint8)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf, valid)
    assert B * K <= 1024
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, 12), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) if valid[k] else ol.DIS_ERR
                      for k in range(K)] for b in range(B)], dtype=np.uint32)
    assert np.array_equal(sc, want), (maxf, K, B)
    wb = np.array([int(np.argmin(w)) if w.min() != ol.DIS_ERR else 0 for w in want])
    assert np.array_equal(res["best_tpl"], wb) and np.array_equal(res["min_dis"], want.min(1))
    eng.close()


@pytest.mark.parametrize("nc,amp", [(13, 3000), (16, 3000), (16, 16383), (15, 32767), (7, 3000)])
def test_dtw_wide_and_narrow_feature_rows_match_oracle(nc, amp):
    """feature rows of the GENERIC front end through the stage-level DTW call: 13..16 coefficients ride the staged kernel's
    16-wide form (48-byte store rows, 32-byte LDS rows), up to 11 its 12-wide form with zero padding, full-scale stores
    (coefficients beyond +-16383 do not fit the -2*coef rows) the generic walk; 1- and 2-frame records, every length gate,
    ties (two coefficients only) in half of the records; batch kernels, one workgroup per pair, automatic"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(nc * 100 + amp % 97)
    maxf, K, B = 90, 50, 64
    orc = ol.Oracle(max_frames=maxf, n_coef=nc)
    eng = Engine(max_frames=maxf, device=0, n_coef=nc)
    tf = rng.integers(1, maxf + 1, K).astype(np.uint32)
    inf = rng.integers(1, maxf + 1, B).astype(np.uint32)
    tf[:3], inf[:3] = [1, 2, maxf], [maxf, 1, 2]
    tm = rng.integers(-amp, amp + 1, (K, maxf + 1, nc)).astype(np.int16)
    im = rng.integers(-amp, amp + 1, (B, maxf, nc)).astype(np.int16)
    tm[::2, :, 2:] = 0
    tm[::2, :, :2] = rng.integers(0, 9, (len(tm[::2]), maxf + 1, 2))
    im[::2, :, 2:] = 0
    im[::2, :, :2] = rng.integers(0, 9, (len(im[::2]), maxf, 2))
    valid = (rng.random(K) > 0.1).astype(np.uint8)
    eng.set_templates_dense(tm, tf, valid)
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, nc), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) if valid[k] else ol.DIS_ERR
                      for k in range(K)] for b in range(B)], dtype=np.uint32)
    assert np.array_equal(sc, want), nc
    assert (want != ol.DIS_ERR).sum() > 600 and np.array_equal(res["min_dis"], want.min(1))
    eng.close()


def test_small_launch_at_the_benchmark_shapes():
    """a few captures of 150..320 frames against 100 templates of 192..320 frames under a 320-frame cap (BASELINE configs[2]'s
    shapes): dtw_limit's band of such a pair (~25 000 points) just fits a workgroup's LDS, so the automatic mode scores it
    with k_dtw_cells; the longest captures against the longest templates do not fit and take its literal walk.  Same scores
    and records as the batch kernels, and the call is at least 1.5x faster for one capture (measured 94 vs 286 us)."""
    import time
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import vad_from_torch
    dev = torch.device("cuda", 0)
    eng = Engine(max_frames=320, device=0)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(2026)
    K = 100
    tfr = rng.integers(192, 321, K)
    tfr[:2] = [320, 192]
    S = synth.buf_len_for(320)
    tp = synth.make_utterances(np.arange(K) % 25, tfr, seed=77, bank=bank, S=S, device=dev)
    tvad, tmf = eng.features_dev(tp)
    torch.cuda.synchronize()
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
    eng.set_templates_dense(tm, vad_from_torch(tvad)["frm_num"].astype(np.uint32))
    fr = [256, 150, 320, 300, 97, 256]
    pcm = synth.make_utterances(rng.integers(0, 25, len(fr)), fr, seed=5, bank=bank, S=S, device=dev)
    med = {}
    for nb in (1, len(fr)):
        x = pcm[:nb].contiguous()
        got = {}
        for mode in (1, 0, 2, 3):
            eng.set_small_launch(mode)
            o = eng.alloc_outputs(nb, dev, mfcc=True, vad=True)
            ts = []
            for i in range(12 if nb == 1 else 2):
                t0 = time.perf_counter()
                eng.recognize_dev(x, o)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            got[mode] = (o["results"].cpu().numpy().copy(), o["scores"].cpu().numpy().copy())
            med[(nb, mode)] = float(np.median(ts[2:])) if nb == 1 else 0.0
        for mode in (0, 2, 3):
            assert np.array_equal(got[mode][0], got[1][0]) and np.array_equal(got[mode][1], got[1][1]), (nb, mode)
        assert (got[1][0].view(np.uint32).reshape(nb, 4)[:, 3] == 0).all()
    assert med[(1, 0)] * 1.5 < med[(1, 1)], med
    eng.close()


def test_mid_sized_launch_takes_the_four_lane_form():
    """256 captures against an 80-slot store at the firmware's shapes (20 480 pairs: far too many for one workgroup per pair,
    far too few to fill the chip with one lane per pair): the automatic mode must pick k_dtw_quad, which shows in the DTW
    kernel's own duration (hipEvents) -- at least 1.4x below the batch kernel's (measured 126 us for it at every size from
    5 000 to 160 000 pairs) -- with identical scores and records, also for ragged lengths, gated-out and erased slots"""
    from stm32_speech_recognition_amd import Engine
    eng = Engine(max_frames=119, device=0)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(14)
    tfr = rng.integers(70, 120, 80)
    tfr[7] = 30                                     # outside the 1/2..2x gate for most captures
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(80) % 25, tfr, seed=8, bank=bank, S=16000))
    store, st = eng.train_store(tp, np.arange(80), n_slots=80)
    store[11 * 4096:11 * 4096 + 2] = 0xFF           # an erased slot
    eng.set_templates_store(store)
    B = 256
    fr = rng.integers(60, 119, B)
    fr[:192] = 110
    d = synth.make_utterances(rng.integers(0, 25, B), fr, seed=9, bank=bank, S=16000, device=torch.device("cuda", 0))
    d[5] = 2048                                     # a silent capture: VAD fail, every score dis_err
    got, dtw_us = {}, {}
    for mode in (1, 3, 0):
        eng.set_small_launch(mode)
        o = eng.alloc_outputs(B, "cuda:0", mfcc=True, vad=True)
        for _ in range(3):
            eng.recognize_dev(d, o)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        for _ in range(10):
            eng.recognize_dev(d, o)
        torch.cuda.synchronize()
        dtw_us[mode] = eng.stage_ms()["dtw"] * 1e3
        eng.set_profiling(False)
        got[mode] = (o["results"].cpu().numpy().copy(), o["scores"].cpu().numpy().copy())
    eng.set_small_launch(0)
    eng.close()
    for mode in (3, 0):
        assert np.array_equal(got[mode][0], got[1][0]) and np.array_equal(got[mode][1], got[1][1]), mode
    sc = got[1][1].view(np.uint32)
    assert (sc[:, 11] == ol.DIS_ERR).all() and (sc[5] == ol.DIS_ERR).all() and (sc[:192, 7] == ol.DIS_ERR).all()
    print("DTW kernel us per launch (batch / four lanes per pair / automatic):", dtw_us)
    assert dtw_us[0] * 1.4 < dtw_us[1] and dtw_us[3] * 1.4 < dtw_us[1], dtw_us


def test_scratch_users_are_ordered_before_host_calls_reuse_the_scratch(golden):
    """an asynchronous sr_recognize_batch_dev that was handed no buffers for its intermediates works in the
    engine's scratch (s_vad, s_mfcc) on the CALLER's stream; a small host-buffer call right behind it runs on an internal
    non-blocking stream and reuses the same scratch.  The engine now marks the end of such a call with an event and orders
    every host-buffer entry point behind it.  Here: a 12 288-capture call on a non-blocking side stream (a few milliseconds
    of kernels) with NO mfcc / vad buffers, directly followed -- no synchronisation -- by one-capture host calls (pinned path)
    and a blocking-copy host call; all three must give what they give alone."""
    from stm32_speech_recognition_amd import Engine
    eng = Engine(max_frames=119, device=0)
    eng.set_templates_store(golden["store"])
    g = golden["pcm"]
    dev = torch.device("cuda", 0)
    big = torch.from_numpy(np.tile(g, (1024, 1)).view(np.int16)).to(dev)          # 12 288 captures
    ref_big = eng.recognize_dev(big, eng.alloc_outputs(len(big), dev, mfcc=False, vad=False))
    torch.cuda.synchronize()
    want_sc, want_res = ref_big["scores"].clone(), ref_big["results"].clone()
    want_small = eng.recognize(g[5:6])
    side = torch.cuda.Stream(device=dev)
    for it in range(6):
        out = eng.alloc_outputs(len(big), dev, mfcc=False, vad=False)
        out["scores"].zero_()
        torch.cuda.synchronize()
        eng.recognize_dev(big, out, stream=side.cuda_stream)                        # asynchronous, scratch intermediates
        if it % 2 == 0:
            small = eng.recognize(g[5:6])                                           # pinned small-call path, internal stream
        else:
            small = eng.recognize(np.tile(g[5:6], (300, 1)), want_mfcc=False, want_vad=False)  # > 256 captures: blocking-copy path
        torch.cuda.synchronize()
        assert torch.equal(out["scores"], want_sc) and torch.equal(out["results"], want_res), it
        assert np.array_equal(small["scores"][0], want_small["scores"][0]) and small["results"][0] == want_small["results"][0], it
    eng.close()


def test_small_launch_soak():
    """thousands of small calls in the automatic mode against one run of the batch kernels: random sub-batches of 1-24 captures
    of random lengths, cut into chunks of 1 / 2 / 5 captures on three streams (several k_dtw_cells launches and their finished-
    pair counters in flight at once), then single captures through the pinned host path -- every score and record identical"""
    from stm32_speech_recognition_amd import Engine
    eng = Engine(max_frames=119, device=0)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(11)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(80) % 25, rng.integers(40, 120, 80), seed=8, bank=bank, S=16000))
    store, st = eng.train_store(tp, np.arange(80), n_slots=80)
    eng.set_templates_store(store)
    n = 96
    dpcm = synth.make_utterances(rng.integers(0, 25, n), rng.integers(30, 119, n), seed=9, bank=bank, S=16000,
                                 device=torch.device("cuda", 0))
    hpcm = dpcm.cpu().numpy().view(np.uint16)
    eng.set_small_launch(1)
    o = eng.alloc_outputs(n, "cuda:0", mfcc=True, vad=True)
    eng.recognize_dev(dpcm, o)
    torch.cuda.synchronize()
    exp_res, exp_sc = o["results"].cpu().numpy().copy(), o["scores"].cpu().numpy().copy()
    eng.set_small_launch(0)
    oo = None
    for it in range(1500):
        b0 = int(rng.integers(0, n - 1))
        nb = int(rng.integers(1, min(24, n - b0) + 1))
        eng.set_pipeline(streams=3, min_chunk=int(rng.choice([1, 2, 5, 4096])), max_chunks=12)
        if oo is None or oo["results"].shape[0] != nb:
            oo = eng.alloc_outputs(nb, "cuda:0", mfcc=True, vad=True)
        eng.recognize_dev(dpcm[b0:b0 + nb].contiguous(), oo)
        torch.cuda.synchronize()
        assert np.array_equal(oo["results"].cpu().numpy(), exp_res[b0:b0 + nb]), (it, b0, nb)
        assert np.array_equal(oo["scores"].cpu().numpy(), exp_sc[b0:b0 + nb]), (it, b0, nb)
    er = exp_res.view(np.uint32).reshape(n, 4)
    for it in range(1500):
        b = int(rng.integers(0, n))
        r = eng.recognize(hpcm[b:b + 1], want_scores=False, want_mfcc=False, want_vad=False)["results"]
        assert (r["best_tpl"][0], r["min_dis"][0], r["frm_num"][0], r["status"][0]) == tuple(er[b]), (it, b)
    eng.close()


def test_concurrent_device_and_host_calls_soak():
    """Two calls of one engine in flight at once, for ~15 s: a device call of 1 229-2 048 captures with the caller's own
    buffers on a side stream (one launch of each batch kernel, every frame-kernel workgroup lives for exactly one work
    item) and, not synchronised with it, a one-capture host call on the engine's internal stream (k_vad_wide, k_mfcc<1>,
    k_dtw_cells).  Every score, record, MFCC row and VAD record of the device call must be what the batch gives alone.
    Round 5: this arrangement found a missing barrier in k_mfcc (the filterbank multipliers in LDS were written by one wave
    AFTER the workgroup's last barrier; a wave that reached its first frame's filterbank before them read what the
    previous workgroup had left there -- the same values as long as that was another k_mfcc workgroup, anything once the
    concurrent call's kernels had used the CU): one wrong MFCC row per ~6 000 calls."""
    import time
    from stm32_speech_recognition_amd import Engine
    eng = Engine(max_frames=119, device=0)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(12)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(80) % 25, rng.integers(40, 120, 80), seed=8, bank=bank, S=16000))
    store, st = eng.train_store(tp, np.arange(80), n_slots=80)
    eng.set_templates_store(store)
    n = 2048
    dev = torch.device("cuda", 0)
    dpcm = synth.make_utterances(rng.integers(0, 25, n), rng.integers(30, 119, n), seed=9, bank=bank, S=16000, device=dev)
    hpcm = dpcm.cpu().numpy().view(np.uint16)
    eng.set_small_launch(1)
    o = eng.recognize_dev(dpcm, eng.alloc_outputs(n, dev, mfcc=True, vad=True))
    torch.cuda.synchronize()
    want = {k: o[k].clone() for k in ("results", "scores", "mfcc", "vad")}
    er = want["results"].cpu().numpy().view(np.uint32).reshape(n, 4)
    eng.set_small_launch(0)
    eng.set_pipeline(streams=3, min_chunk=4096, max_chunks=12)
    side = torch.cuda.Stream(device=dev)
    budget = float(os.environ.get("SR_SOAK_SECONDS", "20"))
    t0, calls = time.time(), 0
    while time.time() - t0 < budget:
        nb = int(rng.integers(1229, 2049))
        b0 = int(rng.integers(0, n - nb + 1))
        oo = eng.alloc_outputs(nb, dev, mfcc=True, vad=True)
        eng.recognize_dev(dpcm[b0:b0 + nb], oo, stream=side.cuda_stream)
        b = int(rng.integers(0, n))
        r = eng.recognize(hpcm[b:b + 1], want_scores=False, want_mfcc=False, want_vad=False)["results"]
        torch.cuda.synchronize()
        calls += 1
        # (a wrong MFCC row shows in most of its utterance's 80 scores; the rows themselves are only looked at to say where)
        if not (torch.equal(oo["scores"], want["scores"][b0:b0 + nb]) and torch.equal(oo["results"], want["results"][b0:b0 + nb])):
            where = {}
            for k in ("vad", "mfcc", "scores", "results"):
                d = (oo[k].reshape(nb, -1) != want[k][b0:b0 + nb].reshape(nb, -1)).any(1)
                where[k] = torch.nonzero(d).ravel()[:4].tolist()
            raise AssertionError(f"call {calls}: the device call differs at utterances {where} (nb {nb}, b0 {b0})")
        assert (r["best_tpl"][0], r["min_dis"][0], r["frm_num"][0], r["status"][0]) == tuple(er[b]), (calls, b)
    assert calls > 200, calls
    eng.close()


def _soak_setup(ext=False, n=2048):
    """engine + 80-slot store + n captures resident in HBM + the outputs of ONE quiet call (the batch kernels, nothing else on
    the chip): what every later call of a soak must reproduce byte for byte"""
    from stm32_speech_recognition_amd import Engine
    rate, cfg, S = (2, dict(fs=16000, nfft=512, n_mel=40), 32000) if ext else (1, {}, 16000)
    eng = Engine(max_frames=119, device=0, **cfg)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(12 + ext)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(80) % 25, rng.integers(40, 120, 80), seed=8, bank=bank, S=S, rate=rate))
    store, st = eng.train_store(tp, np.arange(80), n_slots=80)
    assert (st == 0).all()
    eng.set_templates_store(store)
    dev = torch.device("cuda", 0)
    dpcm = synth.make_utterances(rng.integers(0, 25, n), rng.integers(30, 119, n), seed=9, bank=bank, S=S, device=dev, rate=rate)
    eng.set_small_launch(1)
    o = eng.recognize_dev(dpcm, eng.alloc_outputs(n, dev, mfcc=True, vad=True))
    torch.cuda.synchronize()
    want = {k: o[k].clone() for k in ("results", "scores", "mfcc", "vad")}
    eng.set_small_launch(0)
    return eng, dpcm, want, rng


def _soak_compare(oo, want, b0, nb, what):
    if all(torch.equal(oo[k].reshape(nb, -1), want[k][b0:b0 + nb].reshape(nb, -1)) for k in ("scores", "results", "mfcc", "vad")):
        return
    where = {}
    for k in ("vad", "mfcc", "scores", "results"):
        d = (oo[k].reshape(nb, -1) != want[k][b0:b0 + nb].reshape(nb, -1)).any(1)
        where[k] = torch.nonzero(d).ravel()[:4].tolist()
    raise AssertionError(f"{what}: differs from the quiet run at utterances {where} (nb {nb}, b0 {b0})")


@pytest.mark.parametrize("front", ["ref", "ext"])
def test_lds_poison_between_calls_changes_nothing(front):
    """Race-class test 1 (round 6).  The local data share of EVERY compute unit is overwritten with a seeded pattern
    (sr_lds_poison: one 160 KiB workgroup per CU) before each call; then a batch-sized call (k_vad, the frame kernel, k_dtw_lds)
    and a call small enough for every small-launch form, in all four small-launch modes, must give the bytes of the quiet
    run.  A kernel that reads LDS it has not written -- round 5's defect: a table filled behind the last barrier -- passes
    every other parity test as long as the CU's previous tenant was a workgroup of the same kernel; after the poison it
    cannot.  Fails on the -DSR_INJECT_LDS_RACE build (profiles/experiments/RESULTS.md, round 6)."""
    import ctypes as C
    eng, dpcm, want, rng = _soak_setup(ext=(front == "ext"), n=1536)
    dev = dpcm.device
    nbytes = C.c_uint32(0)
    n_small = 6
    calls = 0
    for rep in range(3):
        for mode in (1, 2, 3, 0):
            eng.set_small_launch(mode)
            for b0, nb in ((0, 1536), (int(rng.integers(0, 1536 - n_small)), n_small)):
                assert eng.L.sr_lds_poison(eng.h, C.c_uint32(0xC0FFEE + calls), None, C.byref(nbytes)) == 0
                oo = eng.recognize_dev(dpcm[b0:b0 + nb], eng.alloc_outputs(nb, dev, mfcc=True, vad=True))
                torch.cuda.synchronize()
                calls += 1
                _soak_compare(oo, want, b0, nb, f"{front} front end, small-launch mode {mode}, {nb} captures after an LDS poison")
    assert nbytes.value == 160 * 1024, nbytes.value   # the poison really covers a CU's whole LDS
    eng.set_small_launch(0)
    eng.close()


def test_two_device_calls_and_a_host_call_in_flight_soak():
    """Race-class test 2 (round 6): THREE calls of one engine in flight for ~10 s -- two device calls on two caller streams
    (1 229-2 048 and 300-700 captures: batch kernels and, for the second, a frame kernel launch that leaves the chip partly
    empty) and an unsynchronised one-capture host call on the engine's internal stream.  Everything the device calls write must
    equal the quiet run; the host call's record too.  Fails on the -DSR_INJECT_LDS_RACE build."""
    import time
    eng, dpcm, want, rng = _soak_setup()
    n = dpcm.shape[0]
    dev = dpcm.device
    hpcm = dpcm.cpu().numpy().view(np.uint16)
    er = want["results"].cpu().numpy().view(np.uint32).reshape(n, 4)
    eng.set_pipeline(streams=3, min_chunk=4096, max_chunks=12)
    s1, s2, s3 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    hog_a, hog_b = torch.empty(96 << 20, dtype=torch.int32, device=dev), torch.empty(96 << 20, dtype=torch.int32, device=dev)
    budget = float(os.environ.get("SR_SOAK_SECONDS", "10"))
    t0, calls = time.time(), 0
    while time.time() - t0 < budget:
        na, nb = int(rng.integers(1229, 2049)), int(rng.integers(300, 701))
        a0, b0 = int(rng.integers(0, n - na + 1)), int(rng.integers(0, n - nb + 1))
        oa, ob = eng.alloc_outputs(na, dev, mfcc=True, vad=True), eng.alloc_outputs(nb, dev, mfcc=True, vad=True)
        if calls & 1:  # every other round under HBM pressure (a 384 MB copy on a third stream: table loads of the kernels get slow)
            with torch.cuda.stream(s3):
                hog_b.copy_(hog_a, non_blocking=True)
        eng.recognize_dev(dpcm[a0:a0 + na], oa, stream=s1.cuda_stream)
        eng.recognize_dev(dpcm[b0:b0 + nb], ob, stream=s2.cuda_stream)
        b = int(rng.integers(0, n))
        r = eng.recognize(hpcm[b:b + 1], want_scores=False, want_mfcc=False, want_vad=False)["results"]
        torch.cuda.synchronize()
        calls += 1
        _soak_compare(oa, want, a0, na, f"call {calls}, first device call")
        _soak_compare(ob, want, b0, nb, f"call {calls}, second device call")
        assert (r["best_tpl"][0], r["min_dis"][0], r["frm_num"][0], r["status"][0]) == tuple(er[b]), (calls, b)
    assert calls > 100, calls
    eng.close()


def test_concurrent_calls_soak_16k_front_end():
    """Race-class test 3 (round 6): the two-calls-in-flight soak of the reference front end above, on the 16 kHz / 512-point /
    40-Mel EXTENSION front end (k_vad<320,160>, k_mfcc_ext; no reference counterpart -- the comparison is with the engine's own
    quiet run), ~10 s.  Fails on the -DSR_INJECT_LDS_RACE build."""
    import time
    eng, dpcm, want, rng = _soak_setup(ext=True, n=1536)
    n = dpcm.shape[0]
    dev = dpcm.device
    hpcm = dpcm.cpu().numpy().view(np.uint16)
    er = want["results"].cpu().numpy().view(np.uint32).reshape(n, 4)
    side = torch.cuda.Stream(device=dev)
    budget = float(os.environ.get("SR_SOAK_SECONDS", "10"))
    t0, calls = time.time(), 0
    while time.time() - t0 < budget:
        nb = int(rng.integers(900, n + 1))
        b0 = int(rng.integers(0, n - nb + 1))
        oo = eng.alloc_outputs(nb, dev, mfcc=True, vad=True)
        eng.recognize_dev(dpcm[b0:b0 + nb], oo, stream=side.cuda_stream)
        b = int(rng.integers(0, n))
        r = eng.recognize(hpcm[b:b + 1], want_scores=False, want_mfcc=False, want_vad=False)["results"]
        torch.cuda.synchronize()
        calls += 1
        _soak_compare(oo, want, b0, nb, f"call {calls}")
        assert (r["best_tpl"][0], r["min_dis"][0], r["frm_num"][0], r["status"][0]) == tuple(er[b]), (calls, b)
    assert calls > 100, calls
    eng.close()


def test_dtw_limit_symbol_exhaustive_against_reference_object():
    """dtw_limit (DTW.C:76-109) through the C ABI against the REFERENCE OBJECT's own dtw_limit (oracle/_ref/libsr_ref.so exports
    it), for 64 random (in, mdl) length pairs -- the reference's dtw() call sets its file statics X1 / X2 / in / mdl
    (DTW.C:65-68, 129-142), the library gets the same lengths -- and EVERY point x, y <= 121 (14 884 per pair, one launch each
    through sr_dtw_limit_batch = the kernel the scalar symbol launches); the scalar symbol itself on 40 random points of
    eight of the pairs after the library's own dtw() call.  Includes pairs the length gate rejects: dtw() returns before
    X1 / X2 are updated (DTW.C:133-137), so dtw_limit keeps answering for the PREVIOUS pair's X1 / X2 with the new lengths."""
    import ctypes as C
    from stm32_speech_recognition_amd import Engine, compat
    if not ol.RefLib.available():
        pytest.fail("oracle/_ref/libsr_ref.so is missing: build it where /root/reference exists (make -C oracle)")
    ref = ol.RefLib()
    ref.L.dtw_limit.restype = C.c_uint8
    eng = Engine(max_frames=119, device=0)
    rng = np.random.default_rng(76109)
    xs, ys = np.meshgrid(np.arange(122, dtype=np.uint16), np.arange(122, dtype=np.uint16), indexing="ij")
    xy = np.ascontiguousarray(np.stack([xs.ravel(), ys.ravel()], 1))
    rows = rng.integers(-300, 300, (2, 120, 12)).astype(np.int16)
    lens = [(int(a), int(b)) for a, b in rng.integers(1, 120, (56, 2))] + [(1, 1), (1, 2), (2, 1), (119, 119), (119, 60), (60, 119),
                                                                           (50, 60), (100, 51)]
    n_gate = 0
    for i, (n_in, n_mdl) in enumerate(lens):
        gated = n_in > 2 * n_mdl or 2 * n_in < n_mdl
        n_gate += gated
        fa, fb = ref.make_ftr(rows[0], n_in), ref.make_ftr(rows[1], n_mdl)
        ref.dtw(fa, fb)
        want = np.fromiter((ref.L.dtw_limit(C.c_uint16(int(x)), C.c_uint16(int(y))) for x, y in xy), np.uint8, len(xy))
        if gated:
            continue   # the statics are a mixture of two calls there; covered through the scalar symbol below
        got = np.zeros(len(xy), np.uint8)
        eng._check(eng.L.sr_dtw_limit_batch(eng.h, xy.ctypes.data_as(C.c_void_p), C.c_uint32(len(xy)), C.c_uint32(n_in),
                                            C.c_uint32(n_mdl), got.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(got, want), (n_in, n_mdl, xy[np.nonzero(got != want)[0][:5]])
    assert n_gate >= 4
    # the scalar symbol after the library's own dtw(), same call sequence on both sides (gated pairs included: stale X1 / X2)
    for (n_in, n_mdl) in [(50, 60)] + lens[:4] + lens[-4:] + [(10, 100), (100, 10), (70, 71)]:
        ref.dtw(ref.make_ftr(rows[0], n_in), ref.make_ftr(rows[1], n_mdl))
        compat.dtw(compat.make_ftr(rows[0], n_in), compat.make_ftr(rows[1], n_mdl))
        for x, y in rng.integers(0, 122, (40, 2)):
            assert compat.dtw_limit(int(x), int(y)) == ref.L.dtw_limit(C.c_uint16(int(x)), C.c_uint16(int(y))), (n_in, n_mdl, x, y)
    eng.close()


def test_get_dis_symbol_against_reference_object():
    """get_dis (DTW.C:45-62) through the C ABI against the REFERENCE OBJECT's get_dis on 10 000 random row pairs at five
    scales, 2 000 full-scale pairs (u32 wrap of the sum of squares), and 4 000 pairs built to land on perfect squares and
    their neighbours (where sqrtf's rounding decides the integer); the scalar symbol itself on 200 of them."""
    import ctypes as C
    from stm32_speech_recognition_amd import Engine, compat
    if not ol.RefLib.available():
        pytest.fail("oracle/_ref/libsr_ref.so is missing: build it where /root/reference exists (make -C oracle)")
    ref = ol.RefLib()
    ref.L.get_dis.restype = C.c_uint32
    rng = np.random.default_rng(4562)
    parts_a, parts_b = [], []
    for scale in (3, 40, 600, 5000, 32767):
        parts_a.append(rng.integers(-scale, scale + 1, (2000, 12)))
        parts_b.append(rng.integers(-scale, scale + 1, (2000, 12)))
    fs = rng.choice(np.array([-32768, 32767, -32767, 0, 1, -1]), (2000, 12))
    parts_a.append(fs)
    parts_b.append(-fs + rng.integers(-1, 2, (2000, 12)))
    # perfect squares: a - b = (k, 0, ..., 0) gives the sum k^2; one more coefficient differing by 1 gives k^2 + 1; and
    # (k, j, 0...) sums that sit just below the next square
    k = rng.integers(1, 65536, 4000)
    a = np.zeros((4000, 12), np.int64)
    b = np.zeros((4000, 12), np.int64)
    a[:, 0] = k // 2
    b[:, 0] = k // 2 - k
    a[1000:2000, 1] = 1
    j = np.sqrt(2 * k[2000:] + 1).astype(np.int64)
    a[2000:3000, 1] = np.clip(j[:1000], 0, 32767)
    a[3000:, 1] = np.clip(j[1000:] - 1, 0, 32767)
    parts_a.append(a)
    parts_b.append(b)
    A = np.ascontiguousarray(np.clip(np.concatenate(parts_a), -32768, 32767).astype(np.int16))
    Bm = np.ascontiguousarray(np.clip(np.concatenate(parts_b), -32768, 32767).astype(np.int16))
    n = len(A)
    want = np.fromiter((ref.L.get_dis(A[i].ctypes.data_as(C.c_void_p), Bm[i].ctypes.data_as(C.c_void_p)) for i in range(n)), np.uint32, n)
    eng = Engine(max_frames=119, device=0)
    got = np.zeros(n, np.uint32)
    eng._check(eng.L.sr_get_dis_batch(eng.h, A.ctypes.data_as(C.c_void_p), Bm.ctypes.data_as(C.c_void_p), C.c_uint32(n),
                                      got.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:8]
    d = (A.astype(np.int64) - Bm.astype(np.int64)) ** 2
    assert (d.sum(1) >= 1 << 32).any()   # the u32 wrap of DTW.C:56-58 is on the path
    assert (np.sqrt(d[12000:13000].sum(1).astype(np.float64)) % 1 == 0).all()   # the perfect squares are perfect squares
    for i in rng.integers(0, n, 200):
        assert compat.get_dis(A[i], Bm[i]) == want[i], i
    eng.close()


def test_small_launch_forms_are_taken(golden):
    """one capture against an 80-slot store at the firmware's shapes: the automatic mode must pick the small-launch kernel forms
    (k_vad_wide, k_mfcc<1>, k_dtw_cells + in-kernel slot scan), which shows as at least 1.5x fewer microseconds per call than
    with them switched off (measured 55 vs 214) -- a silent fallback to the batch kernels would pass every parity test"""
    import time
    from stm32_speech_recognition_amd import Engine
    eng = Engine(max_frames=119, device=0)
    bank = synth.word_bank(25)
    rng = np.random.default_rng(4)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(80) % 25, rng.integers(70, 120, 80), seed=8, bank=bank, S=16000))
    store, st = eng.train_store(tp, np.arange(80), n_slots=80)
    eng.set_templates_store(store)
    d = synth.make_utterances([3], [110], seed=9, bank=bank, S=16000, device=torch.device("cuda", 0))
    o = eng.alloc_outputs(1, "cuda:0", mfcc=False, vad=False)
    med, res = {}, {}
    for mode in (1, 0):
        eng.set_small_launch(mode)
        ts = []
        for i in range(40):
            t0 = time.perf_counter()
            eng.recognize_dev(d, o)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        med[mode] = float(np.median(ts[10:]))
        res[mode] = o["results"].cpu().numpy().copy()
    assert np.array_equal(res[0], res[1])
    assert med[0] * 1.5 < med[1], med
    eng.close()


@pytest.mark.parametrize("K", [513, 700, 1500, 2050])
def test_dtw_large_template_stores_match_oracle(K):
    """more templates than one workgroup of the staged DTW kernel holds (1024 lanes): the length-sorted store is walked in
    equal chunks of at most 512 ranks over the grid's second dimension; erased slots and length-gated pairs included"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(K)
    maxf, B = 48, 9
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(1, maxf, K).astype(np.uint32)
    tm = rng.integers(-2500, 2500, (K, maxf + 1, 12)).astype(np.int16)
    valid = (rng.random(K) > 0.05).astype(np.uint8)
    inf = rng.integers(1, maxf + 1, B).astype(np.uint32)
    im = rng.integers(-2500, 2500, (B, maxf, 12)).astype(np.int16)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf, valid)
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, 12), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) if valid[k] else ol.DIS_ERR
                      for k in range(K)] for b in range(B)], dtype=np.uint32)
    assert np.array_equal(sc, want)
    assert (want != ol.DIS_ERR).sum() > K and np.array_equal(res["min_dis"], want.min(1))
    eng.close()


def test_log_step_table_covers_all_steps(eng119, oracle):
    """MFCC.C:168 on the device = table of step positions built from the host's libm; cross-check the
    device path on filterbank-like magnitudes spanning 0 .. 2^32-1 via constant-spectrum frames is not
    possible directly, so compare whole MFCCs on amplitude sweeps (every decade of pow_spct)."""
    B, nf = 40, 12
    S = 1 + 160 + 80 * (nf - 1) + 7
    rng = np.random.default_rng(8)
    pcm = np.zeros((B, S), np.uint16)
    for b in range(B):
        amp = 2.0 ** (b * 0.4 - 2)
        pcm[b] = np.clip(2048 + rng.normal(0, 1, S) * amp, 0, 65535).astype(np.uint16)
    mid = np.full(B, 2048, np.uint32)
    n, m = eng119.mfcc(pcm, np.ones(B, np.int32), np.full(B, 1 + 160 + 80 * (nf - 1), np.int32), mid)
    for b in range(B):
        a = ol.Atap(2048, 10, 2, 1000)
        n2, m2 = oracle.mfcc(pcm[b], 1, 1 + 160 + 80 * (nf - 1), a)
        assert n[b] == n2 == nf and np.array_equal(m[b, :nf], m2), b


# ----------------------------------------------------------------------------- reference-compatible symbols
@pytest.fixture
def compat_small_launch(request):
    """the implicit engine of the scalar symbols with the small-launch forms on (0: pinned staging, k_vad_wide, ...) or off
    (1: blocking copies, batch kernels)"""
    import ctypes as C
    from stm32_speech_recognition_amd.engine import load_library
    L = load_library()
    L.sr_compat_engine.restype = C.c_void_p
    ce = C.c_void_p(L.sr_compat_engine())
    assert L.sr_set_small_launch(ce, C.c_int(request.param)) == 0
    yield request.param
    L.sr_set_small_launch(ce, C.c_int(0))


@pytest.mark.parametrize("compat_small_launch", [0, 1], indirect=True)
def test_compat_symbols_match_golden(golden, oracle, compat_small_launch):
    from stm32_speech_recognition_amd import compat
    pcm = golden["pcm"][0]
    at = compat.atap_tag()
    compat.noise_atap(pcm, compat.ATAP_LEN, at)
    assert (at.mid_val, at.n_thl, at.z_thl, at.s_thl) == tuple(golden["atap"][0])
    bad = compat.atap_tag(1, 2, 3, 4)
    compat.noise_atap(pcm, 2399, bad)  # VAD.C:33-36: silent no-op
    assert (bad.mid_val, bad.n_thl, bad.z_thl, bad.s_thl) == (1, 2, 3, 4)
    segs = compat.VAD(pcm, compat.VCBUF_LEN, at)
    want = golden["seg"][0]
    for i in range(3):
        assert segs[i] == (None if want[2 * i] < 0 else want[2 * i], None if want[2 * i + 1] < 0 else want[2 * i + 1])
    for name in ("get_mfcc", "GetMfcc", "MFCC_Comp"):
        ftr = compat.get_mfcc(pcm, segs[0][0], segs[0][1], at, name=name)
        n = golden["frm_num"][0]
        assert ftr.frm_num == n
        assert np.array_equal(np.ctypeslib.as_array(ftr.mfcc_dat)[:n * 12].reshape(n, 12), golden["mfcc"][0, :n])
    # fft(): magnitudes of a zero-padded real frame
    frame = (golden["fft_in"][16, :160] & 0xFFFF).astype(np.uint16).view(np.int16)
    fo = compat.fft(frame)
    assert np.array_equal(fo[:512], oracle.fft_mag(frame))
    assert np.array_equal(fo[512:], golden["fft_out"][16, 512:])
    assert compat.fft(np.zeros(1025, np.int16)) is None  # MFCC.C:32-35
    assert np.array_equal(compat.cr4_fft_1024_stm32(golden["fft_in"][3]), golden["fft_out"][3])
    # dtw / get_dis / dtw_limit
    ln, da, db, dd = golden["dtw_len"], golden["dtw_a"], golden["dtw_b"], golden["dtw_dis"]
    for p in range(0, 40):
        fa, fb = compat.make_ftr(da[p], int(ln[p, 0])), compat.make_ftr(db[p], int(ln[p, 1]))
        assert compat.dtw(fa, fb) == dd[p], p
    assert compat.get_dis(da[0, 0], db[0, 0]) == oracle.get_dis(da[0, 0], db[0, 0])
    fa, fb = compat.make_ftr(da[1], 50), compat.make_ftr(db[1], 60)
    compat.dtw(fa, fb)  # leaves X1 = 23, X2 = 26, in = 50, mdl = 60 behind (DTW.C:141-142)
    assert compat.dtw_limit(1, 1) == 0 and compat.dtw_limit(1, 4) == 1 and compat.dtw_limit(10, 2) == 1
    # spch_recg over the flash-layout store
    compat.set_templates(golden["store"])
    for b in range(4):
        label, dis = compat.spch_recg(golden["pcm"][b])
        assert dis == golden["recg_dis"][b]
        idx = int(golden["recg_best"][b]) // 4
        assert label is not None and label[:1] == str(idx).encode()
    label, dis = compat.spch_recg(np.full(16000, 2048, np.uint16))
    assert label is None and dis == compat.DIS_ERR  # main.c:261-266


def test_compat_vad_with_caller_thresholds_outside_the_sample_range(golden, oracle):
    """VAD() takes its thresholds from the caller (VAD.C:97); a mid value that is not a 16-bit quantity -- nothing
    noise_atap could produce -- must still follow the reference's u32 arithmetic (the kernel's two-samples-per-instruction
    |x - mid| path only covers 16-bit mids)."""
    from stm32_speech_recognition_amd import compat
    pcm = golden["pcm"][2]
    x = pcm.astype(np.int64)
    fsum = lambda mid: np.array([np.abs(x[80 * f:80 * f + 160] - mid).sum() for f in range((16000 - 160) // 80)])
    n_found = 0
    for mid, n_thl, z_thl, pct in ((70000, 20, 2000, 0.3), (70000, 20, 2000, 0.55), (65536, 3, 2000, 0.55), (65535, 40, 2000, 0.3),
                                   (100000, 5, 2000, 0.55), (0, 2100, 2, 0.55)):
        fs = fsum(mid)
        s_thl = int(np.sort(fs)[int(len(fs) * pct)])            # a good share of the frames is "loud" by a hair: every sum counts
        at = compat.atap_tag(mid, n_thl, z_thl, s_thl)
        oa = ol.Atap(mid, n_thl, z_thl, s_thl)
        want = oracle.vad(pcm, oa)
        got = compat.VAD(pcm, compat.VCBUF_LEN, at)
        for i in range(3):
            assert got[i] == (None if want[2 * i] < 0 else want[2 * i], None if want[2 * i + 1] < 0 else want[2 * i + 1]), (mid, i)
        n_found += sum(1 for i in range(3) if want[2 * i] >= 0)
    assert n_found >= 5                                          # the cases are not all "no segment"


def test_compat_dtw_slot_scan_uploads_each_model_once(golden):
    """The firmware's slot scan (main.c:279-291: one dtw() per flash slot and utterance) through the scalar symbol:
    a model is uploaded the first time it is seen and one launch scores an input record against every cached model,
    so after the first scan an utterance costs no upload and one launch, and every score still equals the reference objects'."""
    import ctypes as C
    from stm32_speech_recognition_amd import compat
    L = compat._lib()
    st = (C.c_uint32 * 3)()
    L.sr_compat_dtw_stats(st)
    up0, la0 = st[0], st[1]
    ln, da, db, dd = golden["dtw_len"], golden["dtw_a"], golden["dtw_b"], golden["dtw_dis"]
    S, U = 24, 5
    models = [compat.make_ftr(db[p], int(ln[p, 1])) for p in range(S)]
    want = {}
    for rnd in range(2):
        L.sr_compat_dtw_stats(st)
        up1, la1 = st[0], st[1]
        for u in range(U):
            fin = compat.make_ftr(da[u], int(ln[u, 0]))
            for p in range(S):
                d = compat.dtw(fin, models[p])
                if u == p:
                    assert d == dd[p]                   # the pairs the reference objects scored for the fixture
                assert want.setdefault((u, p), d) == d
        L.sr_compat_dtw_stats(st)
        if rnd == 0:                                    # a slot is uploaded when it first passes the length gate (DTW.C:133-137)
            assert st[0] - up0 <= S
        else:                                           # every slot is resident: no upload, one launch per utterance
            assert st[0] == up1 and st[1] - la1 <= U, (st[0] - up1, st[1] - la1)
    # same answers in another order (a pure function of the two records)
    for u in (3, 0):
        fin = compat.make_ftr(da[u], int(ln[u, 0]))
        for p in (7, 23, 0):
            assert compat.dtw(fin, models[p]) == want[(u, p)]
    # a model whose rows past frm_num differ is a different model (DTW.C:152-154 reads them)
    m2 = compat.make_ftr(db[0], int(ln[0, 1]))
    np.ctypeslib.as_array(m2.mfcc_dat)[int(ln[0, 1]) * 12:] = 77
    fin = compat.make_ftr(da[0], int(ln[0, 0]))
    compat.dtw(fin, m2)
    L.sr_compat_dtw_stats(st)
    assert st[2] >= S + 1


def test_delta_mfcc_extension_matches_its_oracle(eng119, oracle, golden):
    """EXTENSION (no reference counterpart): k_delta_mfcc == sr_oracle_delta_mfcc, rows >= frames zero"""
    rng = np.random.default_rng(8)
    B = 40
    mf = np.zeros((B, 119, 12), np.int16)
    fr = np.zeros(B, np.uint32)
    for b in range(B):
        if b < 12 and golden["frm_num"][b] > 0:
            fr[b] = golden["frm_num"][b]
            mf[b] = golden["mfcc"][b]
        else:
            fr[b] = [0, 1, 2, 3, 4, 5, 119][b % 7] if b < 30 else rng.integers(1, 120)
            mf[b, :fr[b]] = rng.integers(-32768, 32768, (fr[b], 12))
    mf[:, 100:] = np.where(np.arange(119)[None, 100:, None] >= fr[:, None, None], 77, mf[:, 100:])  # junk past the record
    got = eng119.delta_mfcc(mf, fr)
    for b in range(B):
        n = int(fr[b])
        assert np.array_equal(got[b, :n], oracle.delta_mfcc(mf[b], n)), b
        assert not got[b, n:].any()


def test_template_store_swap_while_work_is_in_flight(golden):
    """sr_set_templates right behind an asynchronous sr_recognize_batch_dev on a side stream: the call waits for the
    device before it touches the store (the kernels in flight keep reading the old rows), publishes the new store
    atomically, and a failing upload leaves the previous store in place."""
    from stm32_speech_recognition_amd.engine import Engine, SrError, results_from_torch
    dev = torch.device("cuda", 0)
    eng = Engine(device=0)
    store_a = golden["store"].copy()
    store_b = golden["store"].copy()
    nslots = len(store_b) // 4096
    store_b = np.roll(store_b.reshape(nslots, 4096), 3, axis=0).reshape(-1).copy()      # same models, other slots
    pcm = torch.from_numpy(np.tile(golden["pcm"], (64, 1)).view(np.int16)).to(dev)       # enough work to still be running
    B = pcm.shape[0]
    eng.set_templates_store(store_a)
    ref_a = eng.recognize(golden["pcm"], want_mfcc=False, want_vad=False)
    out = eng.alloc_outputs(B, dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    eng.recognize_dev(pcm, out, stream=side.cuda_stream)
    eng.set_templates_store(store_b)                                                     # must not disturb the launch above
    ref_b = eng.recognize(golden["pcm"], want_mfcc=False, want_vad=False)
    torch.cuda.synchronize()
    n = len(golden["pcm"])
    got = out["scores"].cpu().numpy().view(np.uint32)
    for rep in (0, 17, 63):
        assert np.array_equal(got[rep * n:(rep + 1) * n], ref_a["scores"]), rep
    assert np.array_equal(np.roll(ref_a["scores"], 3, axis=1), ref_b["scores"])
    res = results_from_torch(out["results"])
    assert np.array_equal(res["min_dis"][:n], ref_a["results"]["min_dis"])
    with pytest.raises(SrError):                                                         # slot claims more frames than fit
        bad = store_b.copy()
        bad[:4].view(np.uint16)[:] = (12345, 4000)
        eng.set_templates_store(bad)
    again = eng.recognize(golden["pcm"], want_mfcc=False, want_vad=False)
    assert np.array_equal(again["scores"], ref_b["scores"])
    eng.close()


# ----------------------------------------------------------------------------- multi-GPU surface of the C ABI
def _multi_case(devices, golden):
    """sr_multi_* on `devices`: sharded recognition + RCCL all-gather == the single-engine answer, host and device API"""
    from multi_case import multi_case
    multi_case(devices, golden)


@pytest.mark.parametrize("n", [2, 3])
def test_multi_gpu_bookkeeping_with_in_process_collective_double(n):
    """The N > 1 index arithmetic of csrc/sr_multi.cpp (block offsets of the in-place all-gather, padded last shard,
    EMPTY last shard when B < n, read-back from the owning devices / device 0) executed on this 1-GPU box: n ranks on
    device 0 (development hook "multi_allow_dup") over tests/fake_rccl/librccl.so.1 (SR_RCCL_LIBRARY), an in-process stand-in
    that performs the all-gather as stream-ordered device copies between the ranks' buffers.  Own process: the
    collective library is bound once per process, and the other tests of this module use the real RCCL."""
    import json
    import subprocess
    import multi_case as mc
    assert os.path.exists(mc.FAKE_RCCL), "tests/fake_rccl/librccl.so.1 missing: run __graft_entry__.build()"
    env = dict(os.environ, SR_RCCL_LIBRARY=mc.FAKE_RCCL)  # multi_case.py switches the duplicate-device hook on itself
    Bs = [37, 24, n - 1, 1]                 # uneven shards; even shards; B < n (empty last shard); a single capture
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "multi_case.py"), ",".join(["0"] * n),
                        ",".join(map(str, Bs))], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    rep = json.loads([l for l in p.stdout.splitlines() if l.startswith("MULTI_CASE ")][-1][len("MULTI_CASE "):])
    assert [c["B"] for c in rep["cases"]] == Bs and all(c["n"] == n for c in rep["cases"])
    # the double really carried the exchange: every gather is n*(n-1) copies (the own block is in place)
    assert rep["fake_allgathers"] >= 2 * len(Bs) and rep["fake_copies"] == rep["fake_allgathers"] * n * (n - 1), rep


def test_multi_gpu_c_abi_single_device(golden):
    _multi_case([0], golden)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X in one node")
def test_multi_gpu_c_abi_two_devices(golden):
    _multi_case([0, 1], golden)


def _run_bench(extra, env_extra, timeout=900):
    """bench.py as a PLAIN command (no launcher, no WORLD_SIZE / RANK in the environment) -> (the one JSON line, stderr)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                              "SR_BENCH_BACKEND", "SR_BENCH_DEVICE", "SR_RCCL_LIBRARY",
                                                              "SR_BENCH_FORCE_DIST")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=root)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]       # exactly one line on stdout
    return json.loads(lines[0]), p.stderr


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE: the script starts its two ranks itself.  On
    this 1-GPU box the ranks share device 0 and exchange over gloo (test hooks); with two devices the next test runs
    the same command over RCCL."""
    j, _ = _run_bench(["--gpus", "2", "--batch", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                      dict(SR_BENCH_BACKEND="gloo", SR_BENCH_DEVICE="0"))
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["launcher"].startswith("ranks")
    assert j["config"]["batch_per_gpu"] == 4096 and "all-gather" in j["config"]["parallelism"]
    assert abs(j["value"] - 2 * 4096 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6
    assert j["top1_word_accuracy"] == 1.0
    x = j["exchange"]
    assert x["ranks_in_communicator"] == 2 and len(x["step_ms_per_rank"]) == 2 and x["allgather_bytes_per_rank_out"] == 2 * 4096 * 100 * 4
    # strong scaling: the global batch is fixed and split over the ranks
    j, _ = _run_bench(["--gpus", "2", "--scaling", "strong", "--batch", "8192", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                      dict(SR_BENCH_BACKEND="gloo", SR_BENCH_DEVICE="0"))
    assert j["scaling"] == "strong" and j["config"]["batch_per_gpu"] == 4096 and "global batch 8192" in j["config"]["workload"]
    assert abs(j["value"] - 8192 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6


def _check_multi_rank_line(j, n, B, launcher, exchange="scores"):
    assert j["n_gpus"] == n and j["steps"] == 2 and j["launcher"].startswith(launcher)
    assert j["config"]["batch_per_gpu"] == B and "all-gather" in j["config"]["parallelism"]
    assert abs(j["value"] - n * B * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6
    assert j["top1_word_accuracy"] == 1.0
    x = j["exchange"]
    # what one rank receives per step: the whole score matrix (north_star) or, --exchange results, 16 bytes per utterance
    assert x["ranks_in_communicator"] == n and x["allgather_bytes_per_rank_out"] == n * B * (100 * 4 if exchange == "scores" else 16)
    assert ("result records" in j["config"]["parallelism"]) == (exchange == "results")
    r = j["roofline"]  # an N = 8 line carries roofline, cpu_baseline, exchange and a parity flag
    assert 0 < r["frac"] < 1 and 0 < r["timed_step_frac"] < 1 and "valu_frac" in r and "valu_frac_vs_2cycle_peak" in r
    if launcher == "ranks":
        assert "exposed_allgather_ms" in x and "allgather_ms" in x
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["gpu_results_identical_on_sample"] is True
    ps = cb["per_rank_sample"]
    assert ps["ranks"] == n and len(ps["utterances_per_rank"]) == n and min(ps["utterances_per_rank"]) >= 128
    assert ps["rank0_result_records_agree_with_gathered_scan"] is True
    if cb["kind"] == "reference":
        assert cb["port"]["gpu_results_identical_on_sample"] is True and cb["cores_1"]["value"] > 0


def test_bench_eight_ranks_dry_run():
    """The metric is quoted at 1/2/4/8 MI355X and no multi-GPU node is on this side: `python bench.py --gpus 8` exactly as
    the driver types it, eight ranks started by the script itself, here all on device 0 and exchanging over gloo (test
    hooks).  One JSON line with eight per-rank step times, the in-place shard check of measure() on every rank (a failed
    assert there is a non-zero exit code here), the host baseline, and a parity flag that covers a sample of EVERY rank's
    shard read back from the gathered matrix.  Then BASELINE configs[3]'s arithmetic (a fixed global batch split over
    eight ranks, --scaling strong) at a reduced size."""
    hooks = dict(SR_BENCH_BACKEND="gloo", SR_BENCH_DEVICE="0")
    j, _ = _run_bench(["--gpus", "8", "--batch", "2048", "--steps", "2", "--warmup", "1", "--cpu-sample", "256"], hooks, timeout=1500)
    _check_multi_rank_line(j, 8, 2048, "ranks")
    assert len(j["exchange"]["step_ms_per_rank"]) == 8 and j["scaling"] == "weak"
    # the cheaper exchange the driver can A/B against the default (round 6): gather the 16-byte result records instead of the
    # score matrix; parity then reads argmin / distance of every rank's sample from the gathered records
    j, _ = _run_bench(["--gpus", "8", "--batch", "2048", "--steps", "2", "--warmup", "1", "--cpu-sample", "256", "--exchange", "results"],
                      hooks, timeout=1500)
    _check_multi_rank_line(j, 8, 2048, "ranks", exchange="results")
    j, _ = _run_bench(["--gpus", "8", "--scaling", "strong", "--batch", "16384", "--steps", "2", "--warmup", "1", "--cpu-sample", "128"],
                      hooks, timeout=1500)
    _check_multi_rank_line(j, 8, 2048, "ranks")
    assert j["scaling"] == "strong" and "global batch 16384" in j["config"]["workload"]
    assert abs(j["value"] - 16384 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6


def test_bench_eight_devices_single_process_dry_run():
    """`python bench.py --gpus 8 --launcher single`: one process, eight engines (all on device 0 here) and the grouped
    in-place all-gather of csrc/sr_multi.cpp over the in-process RCCL double, with the same self-checks in the line"""
    import multi_case as mc
    j, _ = _run_bench(["--gpus", "8", "--launcher", "single", "--batch", "2048", "--steps", "2", "--warmup", "1", "--cpu-sample", "256"],
                      dict(SR_BENCH_DEVICE="0", SR_RCCL_LIBRARY=mc.FAKE_RCCL), timeout=1500)
    _check_multi_rank_line(j, 8, 2048, "single")
    assert j["exchange"]["every_device_holds_identical_gathered_matrix"] is True


def test_bench_rccl_calls_with_one_rank():
    """The rank-per-GPU path talks to RCCL through torch.distributed (init with device_id, barrier, asynchronous
    all_gather_into_tensor with Work.wait, all_reduce MAX).  Two ranks cannot share a device on RCCL, so on this 1-GPU box
    those very calls run with ONE rank (test hook SR_BENCH_FORCE_DIST): the N > 1 index arithmetic is covered by the gloo
    runs, the RCCL call sequence by this one."""
    j, _ = _run_bench(["--batch", "4096", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"],
                      dict(SR_BENCH_FORCE_DIST="1"))
    assert j["n_gpus"] == 1 and "TEST HOOK" in j["config"]["parallelism"] and j["top1_word_accuracy"] == 1.0
    x = j["exchange"]  # diagnostics of the exchange step: communicator size as RCCL reports it, the collective timed alone
    assert x["backend"] == "nccl" and x["ranks_in_communicator"] == 1 and len(x["step_ms_per_rank"]) == 1
    assert x["allgather_ms"] > 0 and x["exposed_allgather_ms"] >= 0 and x["allgather_bytes_per_rank_out"] == 4096 * 100 * 4


def test_bench_plain_command_single_process_launcher():
    """`python bench.py --gpus 2 --launcher single`: ONE process over the C ABI (sr_multi_create on two "devices",
    sr_multi_recognize_dev per step, one grouped all-gather).  On this 1-GPU box both ranks sit on device 0 and the
    collective library is the in-process double (tests/fake_rccl); with two devices the next test uses RCCL."""
    import multi_case as mc
    j, _ = _run_bench(["--gpus", "2", "--launcher", "single", "--batch", "4096", "--steps", "2", "--warmup", "1"],
                      dict(SR_BENCH_DEVICE="0", SR_RCCL_LIBRARY=mc.FAKE_RCCL))
    assert j["n_gpus"] == 2 and j["launcher"].startswith("single") and "TEST HOOK" in j["config"]["parallelism"]
    assert abs(j["value"] - 2 * 4096 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6
    assert j["top1_word_accuracy"] == 1.0 and j["roofline"]["frac"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X in one node")
@pytest.mark.parametrize("launcher", ["ranks", "single"])
def test_bench_plain_command_over_rccl(launcher):
    """the driver's own line at N = 2 (`python bench.py --gpus 2 ...`, nothing else): one rank per GPU over RCCL, and
    the single-process launcher over the same RCCL"""
    j, _ = _run_bench(["--gpus", "2", "--launcher", launcher, "--batch", "4096", "--steps", "2", "--warmup", "1",
                       "--no-cpu-baseline"], {})
    assert j["n_gpus"] == 2 and j["value"] > 0 and "RCCL" in j["config"]["parallelism"]
    assert "TEST HOOK" not in j["config"]["parallelism"] and j["top1_word_accuracy"] == 1.0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X in one node")
def test_bench_two_ranks_over_rccl():
    """bench.py under torch.distributed.run at N = 2 (WORLD_SIZE set by the launcher): one rank per GPU, backend nccl"""
    import json
    import subprocess
    from bench import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SR_BENCH_BACKEND", None)
    env.pop("SR_BENCH_DEVICE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "RCCL" in line["config"]["parallelism"]


def test_bench_other_configs_and_roofline_keys():
    """the N = 1 line: `roofline` = the dominant kernel alone on the chip (one launch over the whole batch), the chunked
    launches of the timed steps under roofline.overlapped, and configs[1] / configs[4] under other_configs (scaled down
    here), each with a CPU parity check on a sample"""
    j, _ = _run_bench(["--batch", "8192", "--steps", "2", "--warmup", "1", "--cpu-sample", "128", "--other-scale", "16",
                       "--other-steps", "2"], {})
    r = j["roofline"]
    assert r["utterances_per_launch"] == 8192 and r["launches_per_step"] == 1 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["overlapped"]["launches_per_step"] >= 2 and r["overlapped"]["utterances_per_launch"] < 8192
    assert j["cpu_baseline"]["gpu_results_identical_on_sample"] is True
    # the line says what the frame kernel saw: the generator's amplitude and the share of frames per magnitude / filterbank tier
    ws = j["workload_stats"]
    assert ws["gain"] == 1.0 and ws["quiet_frame_fraction"] > 0.9 and ws["frames_counted"] == 32 * 256
    assert abs(ws["quiet_frame_fraction"] + ws["mid_frame_fraction"] + ws["loud_frame_fraction"] - 1) < 1e-9
    assert j["config"]["gain"] == 1.0
    oc = j["other_configs"]
    assert len(oc) == 4 and "configs[4] EXTENSION" in oc[2]["workload"] and "10 templates" in oc[1]["workload"]
    loud = oc[0]  # the metric's configuration at the amplitudes SURVEY.md 8(d) specifies (scaled down here)
    assert "SURVEY.md 8(d)" in loud["workload"] and "100 templates" in loud["workload"] and loud["workload_stats"]["gain"] == 2.4
    assert loud["workload_stats"]["quiet_frame_fraction"] < 0.5 and loud["workload_stats"]["loud_frame_fraction"] > 0.1
    for e in oc[:3]:
        assert e["value"] > 0 and e["parity_on_sample"]["identical"] is True and e["kernel_ms_isolated"]["mfcc"] > 0
        assert e["top1_word_accuracy"] == 1.0
    dp = oc[3]  # the opt-in NON-REFERENCE full-DP scorer, timed alone
    assert "NON-REFERENCE" in dp["workload"] and dp["kernel"] == "k_dtw_dp_band<8>" and dp["parity_on_sample"]["identical"] is True
    assert dp["pairs_per_s"] > 0 and 15000 < dp["cells_per_pair"] < 30000
    lat = j["latency"]  # one call of the drop-in symbols next to the reference's objects on one host core
    assert lat["spch_recg_us"] > 0 and lat["spch_recg_identical"] is True and lat["dtw_identical"] is True
    assert set(lat["sr_recognize_batch_dev_B1_kernel_us"]) == {"vad", "mfcc", "dtw", "argmin", "total"}
    assert j["roofline"]["traffic_stale"] in (True, False) and "sclk" in j


def test_bench_scorer_dp_line():
    """`python bench.py --scorer dp`: the opt-in full-DP scorer timed on its own (never the headline metric), one JSON line
    with the kernel's name, pairs/s, cells/s and a parity sample against its own oracle"""
    j, _ = _run_bench(["--scorer", "dp", "--batch", "2048", "--steps", "2"], {})
    d = j["dp"]
    assert "NON-REFERENCE" in j["metric"] and j["n_gpus"] == 1 and j["value"] == d["value"] > 0
    assert d["kernel"] == "k_dtw_dp_band<8>" and d["parity_on_sample"]["identical"] is True and d["parity_on_sample"]["pairs"] == 6400
    assert abs(d["pairs_per_s"] - 100 * d["value"]) / d["pairs_per_s"] < 1e-9
    j, _ = _run_bench(["--scorer", "dp", "--dp-lanes", "1", "--batch", "512", "--steps", "1"], {})
    assert j["dp"]["kernel"] == "k_dtw_dp_wave64" and j["dp"]["parity_on_sample"]["identical"] is True


# ----------------------------------------------------------------------------- SURVEY 8(f) rows
def test_packed12_host_transport_matches_the_u16_call(eng119, golden):
    """sr_recognize_batch_packed12: 12-bit ADC codes packed two samples in three bytes, unpacked on the device -- identical
    results, scores, MFCC and VAD records to the u16 call on the golden captures (odd and even buffer lengths, a padded row
    stride, and a batch large enough for the chunked upload path)."""
    from stm32_speech_recognition_amd.engine import pack12
    eng119.set_templates_store(golden["store"])
    pcm = golden["pcm"]
    for S in (pcm.shape[1], pcm.shape[1] - 1, pcm.shape[1] - 5):
        p = np.ascontiguousarray(pcm[:, :S])
        want = eng119.recognize(p)
        pk = pack12(p)
        pk = np.concatenate([pk, np.full((len(pk), 7), 0xEE, np.uint8)], 1)  # row stride larger than the payload
        got = eng119.recognize_packed12(pk, S)
        for k in ("results", "scores", "mfcc", "vad"):
            assert np.array_equal(got[k], want[k]), (S, k)
    big = np.tile(pcm, (200, 1))[:2300]  # >= 2048 rows: uploads in chunks that overlap the kernels
    want = eng119.recognize(big, want_mfcc=False, want_vad=False)
    got = eng119.recognize_packed12(pack12(big), big.shape[1], want_mfcc=False, want_vad=False)
    assert np.array_equal(got["results"], want["results"]) and np.array_equal(got["scores"], want["scores"])


def test_recognize_segments_matches_golden(eng119, golden):
    """multi-segment recognition against the reference objects' per-segment get_mfcc + dtw"""
    eng119.set_templates_store(golden["store"])
    res, sc, vd = eng119.recognize_segments(golden["multi_pcm"])
    assert np.array_equal(vd["seg"], golden["multi_seg"])
    for s_ in range(3):
        assert np.array_equal(res[s_]["status"], golden["multi_status"][:, s_]), s_
        assert np.array_equal(res[s_]["frm_num"], golden["multi_frm"][:, s_])
        assert np.array_equal(res[s_]["min_dis"], golden["multi_dis"][:, s_])
        assert np.array_equal(res[s_]["best_tpl"], golden["multi_best"][:, s_])
        assert np.array_equal(sc[s_], golden["multi_scores"][:, s_])


def test_real_speech_matches_golden(eng119):
    """REAL speech on the GPU: capture buffers cut from the reference's own recordings and what the reference's
    compiled objects made of them (tests/golden/real_speech.npz, make_real_golden.py): noise_atap, every VAD
    segment, MFCC of every segment, all DTW scores, argmin -- all segments, bit for bit"""
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "real_speech.npz"))
    eng119.set_templates_store(g["store"])
    res, sc, vd = eng119.recognize_segments(g["pcm"])
    assert np.array_equal(vd["seg"], g["seg"])
    for f, col in (("mid_val", 0), ("n_thl", 1), ("z_thl", 2), ("s_thl", 3)):
        assert np.array_equal(vd[f].astype(np.uint32), g["atap"][:, col]), f
    for s_ in range(3):
        assert np.array_equal(res[s_]["status"], g["status"][:, s_]), s_
        assert np.array_equal(res[s_]["frm_num"], g["frm"][:, s_])
        assert np.array_equal(res[s_]["min_dis"], g["dis"][:, s_])
        assert np.array_equal(res[s_]["best_tpl"], g["best"][:, s_])
        assert np.array_equal(sc[s_], g["scores"][:, s_])
    # MFCC rows of every closed segment through the stage-level API
    n = 0
    for i in range(g["pcm"].shape[0]):
        for s_ in range(3):
            if g["status"][i, s_] == 0:
                fr, m = eng119.mfcc(g["pcm"][i][None, :], np.array([g["seg"][i, 2 * s_]]), np.array([g["seg"][i, 2 * s_ + 1]]),
                                    np.array([g["atap"][i, 0]]))
                assert fr[0] == g["frm"][i, s_] and np.array_equal(m[0, :fr[0]], g["mfcc"][i, s_, :fr[0]]), (i, s_)
                n += 1
    assert n >= 20


def test_train_store_matches_reference_layout(golden):
    """save_mdl for a batch of captures -> flash image (Flash.C:17-67), then recognise with that image"""
    from stm32_speech_recognition_amd import Engine
    from test_oracle import expected_store_image
    e = Engine(max_frames=119, device=0)
    pcm = golden["pcm"]
    B = pcm.shape[0]
    pcm2 = np.concatenate([pcm, np.full((1, pcm.shape[1]), 2048, np.uint16)])  # + one silent capture: VAD_fail
    slots = np.array([3 * i for i in range(B)] + [79], np.uint32)
    store, status = e.train_store(pcm2, slots)
    assert status[B] == 1 and (status[:B] == 0).all()
    want = expected_store_image([golden["mfcc"][b] for b in range(B)] + [None], list(golden["frm_num"]) + [0], slots,
                                status=status)
    assert np.array_equal(store, want)
    assert (store[79 * 4096:80 * 4096] == 0xFF).all()  # failed capture leaves its slot untouched
    # the trained image is a valid template store: every training capture matches itself with distance 0
    e.set_templates_store(store)
    out = e.recognize(pcm)
    assert np.array_equal(out["results"]["best_tpl"], slots[:B]) and (out["results"]["min_dis"] == 0).all()
    e.close()


@pytest.mark.parametrize("lanes", [1, 4, 8, 16])
def test_dtw_dp_extension_matches_its_oracle(lanes):
    """opt-in NON-REFERENCE full-DP scorer against its own CPU definition, every kernel variant (lanes per pair of the
    band-limited wavefront; 1 = one wave per pair); also sanity: D(a,a) = 0.  Lengths 1, 2 and around the strip widths,
    an erased slot, pairs outside the 1/2..2x gate."""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(77)
    maxf, K, B = 150, 10, 40
    orc = ol.Oracle(max_frames=maxf)
    tf = np.array([1, 2, 40, 64, 65, 100, 128, 129, 150, 77], np.uint32)
    tm = rng.integers(-2500, 2500, (K, maxf + 1, 12)).astype(np.int16)
    valid = np.ones(K, np.uint8)
    valid[9] = 0
    inf = rng.integers(1, maxf + 1, B).astype(np.uint32)
    inf[:12] = (64, 65, 128, 150, 1, 2, 3, 4, 7, 8, 9, 17)
    im = rng.integers(-2500, 2500, (B, maxf, 12)).astype(np.int16)
    im[0, :64] = tm[3, :64]  # identical sequence -> score 0
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf, valid)
    eng.set_dp_lanes(lanes)
    sc = eng.dtw_dp(im, inf)
    want = orc.dtw_dp_batch(im, inf, tm, np.where(valid != 0, tf, 0))
    assert np.array_equal(want[3], np.array([orc.dtw_dp(im[3], inf[3], tm[k], tf[k] if valid[k] else 0) for k in range(K)], np.uint32))
    assert np.array_equal(sc, want)
    assert sc[0, 3] == 0 and (sc[:, 9] == ol.DIS_ERR).all()
    assert (want == ol.DIS_ERR).sum() > 10 and (want != ol.DIS_ERR).sum() > 40
    eng.close()


@pytest.mark.parametrize("lanes,seed", [(8, 0), (8, 1), (16, 2), (4, 3), (0, 4)])
def test_dtw_dp_random_lengths_match_its_oracle(lanes, seed):
    """full-DP scorer on 200 x 48 = 9 600 pairs of random lengths (1..maxf, every gate outcome, bands of every shape: the
    strips of a wave carry groups that start, narrow and finish at different steps), random features incl. repeated rows
    (distance-0 plateaus), frame caps that are / are not multiples of the strip width"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(5100 + seed)
    maxf = int(rng.choice([37, 64, 101, 130]))
    if lanes == 0:
        maxf = 400  # automatic choice: three 8-lane workgroups no longer fit a CU's LDS at this cap -> the 16-lane shape
    K, B = 48, 200
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(1, maxf + 1, K).astype(np.uint32)
    tm = rng.integers(-3000, 3000, (K, maxf + 1, 12)).astype(np.int16)
    inf = rng.integers(1, maxf + 1, B).astype(np.uint32)
    im = rng.integers(-3000, 3000, (B, maxf, 12)).astype(np.int16)
    for b in range(0, B, 7):   # plateaus: runs of identical rows, and copies of template stretches
        im[b, 5:5 + maxf // 3] = im[b, 5]
        k = b % K
        n = min(int(inf[b]), int(tf[k]))
        im[b, :n] = tm[k, :n]
    valid = (rng.random(K) > 0.1).astype(np.uint8)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf, valid)
    eng.set_dp_lanes(lanes)
    got = eng.dtw_dp(im, inf)
    want = orc.dtw_dp_batch(im, inf, tm, np.where(valid != 0, tf, 0), n_threads=8)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    assert 0.2 < (want == ol.DIS_ERR).mean() < 0.8
    eng.close()


def test_dtw_dp_full_scale_store_takes_the_generic_kernel():
    """coefficients beyond +-16383 do not fit the band kernel's -2*coef rows: the store is scored by the one-wave-per-pair
    kernel, same results as the oracle (u32 wrap of the squared distance included)"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(78)
    maxf, K, B = 70, 6, 12
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(30, maxf + 1, K).astype(np.uint32)
    tm = rng.integers(-32768, 32768, (K, maxf + 1, 12)).astype(np.int16)
    inf = rng.integers(30, maxf + 1, B).astype(np.uint32)
    im = rng.integers(-32768, 32768, (B, maxf, 12)).astype(np.int16)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf)
    assert np.array_equal(eng.dtw_dp(im, inf), orc.dtw_dp_batch(im, inf, tm, tf))
    eng.close()


@pytest.mark.parametrize("lanes,ragged", [(8, False), (8, True), (16, True), (4, False), (1, False)])
def test_dtw_dp_benchmark_shape_matches_its_oracle(lanes, ragged):
    """the full-DP scorer at the benchmark's shape: features of 256-frame utterances from the real front end (frame cap
    320), 100 templates of 192..320 frames with ~10 % outside the 1/2..2x gate (DTW.C:133-137) and one erased slot,
    96 x 100 = 9 600 pairs; frame counts taken from the VAD records on the device (in_frames == NULL path).  `ragged`:
    utterances of 100..320 frames, so the groups of a wave walk different bands and finish in different strips."""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import vad_from_torch
    rng = np.random.default_rng(91 + lanes + ragged)
    T, maxf, K, B = 256, 320, 100, 96
    orc = ol.Oracle(max_frames=maxf)
    bank = synth.word_bank(10)
    tfr = [int(v) for v in rng.integers(192, 321, K)]
    for k in range(0, K, 10):
        tfr[k] = int(rng.integers(40, 120))  # outside the gate for 256-frame utterances -> dis_err
    tm, tf = _oracle_templates(orc, bank, tfr, seed=17, S=synth.buf_len_for(320))
    valid = np.ones(K, np.uint8)
    valid[7] = 0
    frames = [int(v) for v in rng.integers(100, 321, B)] if ragged else [T] * B
    pcm = synth.make_utterances(rng.integers(0, 10, B), frames, seed=19, bank=bank, S=synth.buf_len_for(320), device="cuda:0")
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf, valid)
    eng.set_dp_lanes(lanes)
    vad, mfcc = eng.features_dev(pcm)
    sc = torch.full((B, K), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
    eng.dtw_dp_dev(mfcc, sc, vad=vad)
    torch.cuda.synchronize()
    vd = vad_from_torch(vad)
    assert (vd["status"] == 0).all() and np.array_equal(vd["frm_num"], np.array(frames, np.uint32))
    got = sc.cpu().numpy().view(np.uint32)
    want = orc.dtw_dp_batch(mfcc.cpu().numpy(), vd["frm_num"], tm, np.where(valid != 0, tf, 0), n_threads=min(32, os.cpu_count() or 8))
    assert np.array_equal(got, want)
    assert (want[:, 7] == ol.DIS_ERR).all() and (want != ol.DIS_ERR).sum() > 0.5 * B * K
    if not ragged:
        assert (want[:, 0::10] == ol.DIS_ERR).all()
    eng.close()


def test_extension_full_shape_matches_its_oracle():
    """EXTENSION front end at BASELINE configs[4]'s per-GPU shape (>= 1024 utterances per config): 16 kHz /
    512-point / 40 Mel, frame cap 320, 256-frame utterances x 500 templates of 192..320 frames with ~10 % outside the
    1/2..2x gate (DTW.C:133-137), B = 1024.  MFCC s16, all 500 scores u32 and the argmin against the parametrised oracle;
    the staged DTW kernel must really run its 7 x 125 geometry with the 16 384-entry tie table (squared distances of this
    front end reach beyond a table that ends at a root of 8 192)."""
    import ctypes as C
    from stm32_speech_recognition_amd import Engine, engine
    from stm32_speech_recognition_amd.engine import results_from_torch, vad_from_torch
    rng = np.random.default_rng(1640)
    T, maxf, K, B = 256, 320, 500, 1024
    geo = (C.c_uint32 * 5)()
    assert engine.load_library().sr_dtw_geometry(C.c_uint32(K), C.c_uint32(maxf), geo) == 0
    assert (geo[0], geo[1], geo[2]) == (7, 125, 16384), list(geo)
    cfg = dict(fs=16000, nfft=512, n_mel=40)
    orc = ol.Oracle(max_frames=maxf, **cfg)
    eng = Engine(max_frames=maxf, device=0, **cfg)
    bank = synth.word_bank(100)
    tfr = rng.integers(192, 321, K)
    out_gate = rng.permutation(K)[:K // 10]
    tfr[out_gate] = rng.integers(60, 128, len(out_gate))  # 2 * frames < 256 -> dis_err for every utterance
    tpcm = synth.make_utterances(np.arange(K) % 100, tfr, seed=77, bank=bank, S=synth.buf_len_for(320, 2), device="cuda:0", rate=2)
    tvad, tmf = eng.features_dev(tpcm)
    torch.cuda.synchronize()
    tv = vad_from_torch(tvad)
    assert (tv["status"] == 0).all() and np.array_equal(tv["frm_num"], tfr)
    tm = np.concatenate([tmf.cpu().numpy(), np.zeros((K, 1, 12), np.int16)], 1)
    # the template features themselves against the oracle (a strided sample: the batch below covers the front end in full)
    tp_host = synth.as_u16_numpy(tpcm)
    for k in range(0, K, 25):
        rc, a = orc.noise_atap(tp_host[k])
        seg = orc.vad(tp_host[k], a)
        n, m = orc.mfcc(tp_host[k], seg[0], seg[1], a)
        assert n == tfr[k] and np.array_equal(m, tm[k, :n]), k
    eng.set_templates_dense(tm, tfr.astype(np.uint32))
    pcm = synth.make_utterances(rng.integers(0, 100, B), [T] * B, seed=1000, bank=bank, S=synth.buf_len_for(T, 2), device="cuda:0", rate=2)
    out = eng.recognize_dev(pcm, eng.alloc_outputs(B, "cuda:0"))
    torch.cuda.synchronize()
    res = results_from_torch(out["results"])
    tpl = orc.make_templates(tm, tfr.astype(np.uint32))
    ores, omf, osc = orc.recognize_batch(synth.as_u16_numpy(pcm), tpl, n_threads=min(64, os.cpu_count() or 8))
    assert (ores["status"] == 0).all() and (ores["frm_num"] == T).all()
    assert np.array_equal(out["mfcc"].cpu().numpy(), omf)
    gsc = out["scores"].cpu().numpy().view(np.uint32)
    assert np.array_equal(gsc, osc)
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(res[f], ores[f]), f
    assert (osc[:, out_gate] == ol.DIS_ERR).all() and (osc != ol.DIS_ERR).sum() == B * (K - len(out_gate))
    eng.close()


def test_extension_front_end_segments_at_odd_and_first_samples():
    """EXTENSION frame kernel with caller-chosen segments (sr_mfcc_batch): a segment that starts at sample 1 -- the pair
    x[-1], x[0] of its first lane lies half outside the capture row, and x[0] is the pre-emphasis predecessor of the
    segment's first sample (MFCC.C:119) --, odd and even starts, against the oracle; plus per-record failure
    (sr_mfcc_batch_status): a segment outside the buffer, one shorter than a frame and one longer than the cap among good ones."""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(5)
    for cfg, fl, hop in ((dict(fs=16000, nfft=512, n_mel=40), 320, 160), ({}, 160, 80)):
        maxf = 40
        orc = ol.Oracle(max_frames=maxf, **cfg)
        eng = Engine(max_frames=maxf, device=0, **cfg)
        S = 16000
        B = 12
        pcm = (2048 + rng.integers(-900, 900, (B, S))).astype(np.uint16)
        starts = np.array([1, 1, 2, 3, 4, 5, 161, 7777, 1, 100, 50, 200], np.int32)
        nfr = np.array([1, 17, 33, 5, 8, 9, 12, 30, 40, 3, 2, 6], np.int32)
        ends = starts + fl + hop * (nfr - 1) + rng.integers(0, hop, B).astype(np.int32)
        mid = rng.integers(1900, 2200, B).astype(np.uint32)
        # bad records among the good ones
        starts[9], ends[9] = 0, 4000            # start < 1: the reference would read before the buffer
        ends[10] = starts[10] + fl - 1          # shorter than a frame
        ends[11] = starts[11] + fl + hop * maxf  # maxf + 1 frames
        n, mf, st = eng.mfcc_status(pcm, starts, ends, mid)
        assert st.tolist() == [0] * 9 + [ol.ST_SEG_OOB, ol.ST_MFCC_FAIL, ol.ST_MFCC_FAIL]
        assert n[9:].tolist() == [0, 0, 0] and not mf[9:].any()
        wrong = []
        for b in range(9):
            a = ol.Atap(int(mid[b]), 0, 0, 0)
            nn, m = orc.mfcc(pcm[b], int(starts[b]), int(ends[b]), a)
            assert nn == n[b] == nfr[b], (b, nn, n[b])
            if not np.array_equal(mf[b, :nn], m) or mf[b, nn:].any():
                wrong.append((b, int(starts[b]), [int(f) for f in np.nonzero((mf[b, :nn] != m).any(1))[0][:6]]))
        assert not wrong, (cfg, wrong)  # (record, start sample, first differing frames)
        eng.close()


@pytest.mark.parametrize("ci", range(len(ol.GENERIC_CONFIGS)))
def test_generic_front_end_matches_oracle(ci):
    """GENERIC front end (round 4): the reference's compile-time constants (MFCC.H:7-16, VAD.H:4-8, ADC.H:7-11) as
    run-time configuration -- other sampling rates, framings, filter counts and feature widths -- through k_mfcc_gen, the
    VAD instance of the framing and (feature rows of 13..16 coefficients) the 16-wide form of k_dtw_lds.  Whole path against the
    parametrised oracle: thresholds, every VAD segment, frame counts, MFCC s16, all scores, argmin; silent and
    over-long captures included.  No reference counterpart for the constants; every arithmetic rule is the reference's."""
    from stm32_speech_recognition_amd import Engine
    ekw, okw = ol.GENERIC_CONFIGS[ci]
    rng = np.random.default_rng(400 + ci)
    maxf, K, B = 150, 14, 72
    orc = ol.Oracle(max_frames=maxf, **okw)
    eng = Engine(max_frames=maxf, device=0, **ekw)
    rate = 2 if ekw.get("fs", 8000) == 16000 else 1
    bank = synth.word_bank(7)
    S = synth.buf_len_for(120, rate) + 8000  # room for the longer noise heads (up to 960 ms)
    nl = orc.noise_len

    def captures(n, lo, hi, seed):
        fr = [int(v) for v in rng.integers(lo, hi, n)]
        p = synth.as_u16_numpy(synth.make_utterances(rng.integers(0, 7, n), fr, seed=seed, bank=bank, rate=rate, S=S))
        if nl > 2400 * rate:  # the generator's quiet head is 300 ms: stretch it with more of the same noise
            pad = (2048 + rng.normal(0, 8, (n, nl - 2400 * rate))).astype(np.uint16)
            p = np.concatenate([pad, p], 1)[:, :S]
        return np.ascontiguousarray(p)

    tp = captures(K, 30, 110, 61)
    nc = orc.n_coef
    tm, tf = np.zeros((K, maxf + 1, nc), np.int16), np.zeros(K, np.uint32)
    for k in range(K):
        rc, a = orc.noise_atap(tp[k])
        seg = orc.vad(tp[k], a)
        assert seg[1] >= 0, k
        n, m = orc.mfcc(tp[k], seg[0], seg[1], a)
        tm[k, :n], tf[k] = m, n
    assert (tf > 0).all() and len(set(tf.tolist())) > 4
    valid = np.ones(K, np.uint8)
    valid[2] = 0
    pcm = captures(B, 20, 125, 62)
    pcm[5] = 2048
    pcm[6] = captures(1, 170, 171, 63)[0] if synth.buf_len_for(170, rate) + 8000 <= S else pcm[6]
    eng.set_templates_dense(tm, tf, valid)
    out = eng.recognize(pcm)
    tpl = orc.make_templates(tm, tf, valid)
    ores, omf, osc = orc.recognize_batch(pcm, tpl, n_threads=8)
    for b in range(B):
        rc, a = orc.noise_atap(pcm[b])
        seg = orc.vad(pcm[b], a)
        v = out["vad"][b]
        assert (v["mid_val"], v["n_thl"], v["z_thl"], v["s_thl"]) == a.astuple(), b
        assert np.array_equal(v["seg"], seg), b
    assert np.array_equal(out["results"]["status"], ores["status"]) and np.array_equal(out["results"]["frm_num"], ores["frm_num"])
    assert out["mfcc"].shape == omf.shape == (B, maxf, nc) and np.array_equal(out["mfcc"], omf)
    assert np.array_equal(out["scores"], osc)
    for f in ("best_tpl", "min_dis"):
        assert np.array_equal(out["results"][f], ores[f]), f
    assert (ores["status"] == 0).sum() > B // 2 and ores["status"][5] == ol.ST_VAD_FAIL and len(set(ores["frm_num"].tolist())) > 10
    assert (osc[:, 2] == ol.DIS_ERR).all() and (osc != ol.DIS_ERR).sum() > B
    # the device-resident entry point and template training run the same kernels
    if nc == 12:
        d = eng.delta_mfcc(out["mfcc"], out["results"]["frm_num"])
        assert d.shape == out["mfcc"].shape
    eng.close()


@pytest.mark.parametrize("seed", range(32))
def test_generic_front_end_random_configurations(seed):
    """GENERIC front end, configurations drawn at random from everything sr_create accepts (sampling rate 4..48 kHz in
    4 kHz steps x the framings that give a 160 / 240 / 256 / 320 / 400 / 512-sample frame, any even filter count 4..64, any
    feature width 1..16, the shortest noise head that holds whole frames and whole 30 ms blocks): engine == parametrised
    oracle on thresholds, segments, frame counts, MFCC, scores and argmin, whatever the VAD makes of the captures."""
    from math import gcd
    from stm32_speech_recognition_amd import Engine
    # a configuration whose captures give the VAD no usable template segment is RE-DRAWN (a skipped seed would be an
    # untested configuration): every one of the 32 seeds runs
    for attempt in range(16):
        rng = np.random.default_rng(9000 + seed + 1000 * attempt)
        combos = [(fs, ft) for fs in range(4000, 48001, 4000) for ft in range(2, 130, 2)
                  if fs // 1000 * ft in (160, 240, 256, 320, 400, 512)]
        fs, ft = combos[int(rng.integers(0, len(combos)))]
        n_mel, n_coef = int(rng.integers(2, 33)) * 2, int(rng.integers(1, 17))
        if (fs, ft, n_mel, n_coef) == (8000, 20, 24, 12):
            n_mel = 26
        noise_ms = 30 * ft // gcd(30, ft)                      # whole 30 ms blocks (VAD.C:48-63) and whole frames
        noise_ms *= max(1, -(-240 // noise_ms))                # at least 240 ms of noise
        ekw = dict(fs=fs, frame_time_ms=ft, frame_mov_ms=ft // 2, n_mel=n_mel, n_coef=n_coef, noise_len_ms=noise_ms)
        okw = dict(fs=fs, frame_time=ft, frame_mov_t=ft // 2, n_mel=n_mel, n_coef=n_coef, noise_len_t=noise_ms)
        maxf, K, B = 120, 9, 40
        orc = ol.Oracle(max_frames=maxf, **okw)
        eng = Engine(max_frames=maxf, device=0, **ekw)
        fl, hop, nl = orc.frame_len, orc.hop, orc.noise_len
        # captures: quiet head of nl samples, then bursts of a few tones with pauses, 12-bit codes
        S = (nl + hop * 150 + fl + 7) // 8 * 8

        def captures(n):
            t = np.arange(S)[None, :]
            f0 = rng.uniform(0.002, 0.2, (n, 1))
            env = (np.sin(2 * np.pi * t / (hop * rng.uniform(20, 90, (n, 1))) + rng.uniform(0, 6, (n, 1))) > rng.uniform(-0.3, 0.6, (n, 1)))
            sig = 600 * np.sin(2 * np.pi * f0 * t) * env + 250 * np.sin(2 * np.pi * 3.1 * f0 * t) * env
            sig[:, :nl + hop * 4] = 0
            return np.clip(2048 + sig + rng.normal(0, 7, (n, S)), 0, 4095).astype(np.uint16)

        tp = captures(3 * K)
        tm, tf = np.zeros((K, maxf + 1, n_coef), np.int16), np.zeros(K, np.uint32)
        k = 0
        for row in tp:
            rc, a = orc.noise_atap(row)
            seg = orc.vad(row, a)
            if seg[1] < 0 or seg[0] < 1:
                continue
            n, m = orc.mfcc(row, seg[0], seg[1], a)
            if n == 0:
                continue
            tm[k, :n], tf[k] = m, n
            k += 1
            if k == K:
                break
        if k >= 2:
            break
        eng.close()
    assert k >= 2, f"16 draws without a usable template segment (last: {ekw})"
    tm, tf = tm[:k], tf[:k]
    pcm = captures(B)
    pcm[3] = 2048
    eng.set_templates_dense(tm, tf)
    if seed & 1:  # the chunked multi-stream pipeline with ragged chunks, feature records n_coef wide
        eng.set_pipeline(streams=3, min_chunk=7, max_chunks=12)
    out = eng.recognize(pcm)
    tpl = orc.make_templates(tm, tf)
    ores, omf, osc = orc.recognize_batch(pcm, tpl, n_threads=8)
    for b in range(B):
        rc, a = orc.noise_atap(pcm[b])
        seg = orc.vad(pcm[b], a)
        v = out["vad"][b]
        assert (v["mid_val"], v["n_thl"], v["z_thl"], v["s_thl"]) == a.astuple(), (ekw, b)
        assert np.array_equal(v["seg"], seg), (ekw, b)
    assert np.array_equal(out["results"]["status"], ores["status"]) and np.array_equal(out["results"]["frm_num"], ores["frm_num"]), ekw
    assert np.array_equal(out["mfcc"], omf), ekw
    assert np.array_equal(out["scores"], osc), ekw
    for f in ("best_tpl", "min_dis"):
        assert np.array_equal(out["results"][f], ores[f]), (ekw, f)
    for mode in (2, 3, 1, 0):  # one workgroup per pair (k_dtw_cells, any feature width up to 16) / four lanes per pair / never / automatic
        eng.set_small_launch(mode)
        out2 = eng.recognize(pcm)
        assert np.array_equal(out2["scores"], osc) and out2["results"].tobytes() == out["results"].tobytes(), (ekw, mode)
    # every VAD segment matched like segment 0 (sr_recognize_segments_batch), and template training into a slot image
    # whose records are n_coef wide (sr_train_store), on the same kernels
    sres, ssc, svd = eng.recognize_segments(pcm[:12])
    for b in range(12):
        wr, ws = orc.recognize_segments(pcm[b], tpl)
        for s_ in range(3):
            assert (sres[s_][b]["status"], sres[s_][b]["frm_num"], sres[s_][b]["min_dis"], sres[s_][b]["best_tpl"]) == \
                   (wr[s_]["status"], wr[s_]["frm_num"], wr[s_]["min_dis"], wr[s_]["best_tpl"]), (ekw, b, s_)
            assert np.array_equal(ssc[s_][b], ws[s_]), (ekw, b, s_)
    stride = 4 + 2 * n_coef * (maxf + 1)
    store, st = eng.train_store(tp[:6], np.arange(6), n_slots=6, stride=stride)
    for i in range(6):
        rc, a = orc.noise_atap(tp[i])
        seg = orc.vad(tp[i], a)
        n, m = (0, None) if seg[1] < 0 or seg[0] < 1 else orc.mfcc(tp[i], seg[0], seg[1], a)
        slot = store[i * stride:(i + 1) * stride]
        if n:
            assert st[i] == 0 and tuple(slot[:4].view(np.uint16)) == (12345, n), (ekw, i)
            assert np.array_equal(slot[4:4 + n * n_coef * 2].view(np.int16).reshape(n, n_coef), m), (ekw, i)
        else:
            assert st[i] != 0 and (slot == 0xFF).all(), (ekw, i)
    eng.close()


@pytest.mark.parametrize("which", ["ref", "ext", "gen"])
def test_mfcc_status_random_segments(which):
    """sr_mfcc_batch_status on 96 random segments per front end: any start (odd, even, 1), lengths from below one frame to
    beyond the cap, ends beyond the buffer, reversed bounds -- good records equal the oracle's get_mfcc, bad ones fail
    alone (frm_num 0, zero record, status) without disturbing their neighbours"""
    from stm32_speech_recognition_amd import Engine
    ekw, okw = {"ref": ({}, {}), "ext": (dict(fs=16000, nfft=512, n_mel=40), dict(fs=16000, nfft=512, n_mel=40)),
                "gen": ol.GENERIC_CONFIGS[3]}[which]
    rng = np.random.default_rng({"ref": 1, "ext": 2, "gen": 3}[which])
    maxf, B, S = 48, 96, 12000
    orc = ol.Oracle(max_frames=maxf, **okw)
    eng = Engine(max_frames=maxf, device=0, **ekw)
    fl, hop, nc = orc.frame_len, orc.hop, orc.n_coef
    pcm = np.clip(2048 + rng.normal(0, 500, (B, S)), 0, 4095).astype(np.uint16)
    starts = rng.integers(1, S - fl, B).astype(np.int32)
    starts[:6] = (1, 1, 2, 3, 4, 5)
    length = rng.integers(fl, fl + hop * (maxf - 1) + hop, B)
    ends = np.minimum(starts + length, S).astype(np.int32)
    kind = rng.integers(0, 10, B)
    kind[:6] = 9
    ends = np.where(kind == 0, starts + rng.integers(0, fl, B), ends).astype(np.int32)            # shorter than a frame
    ends = np.where(kind == 1, starts + fl + hop * maxf + rng.integers(0, 3 * hop, B), ends)      # beyond the frame cap
    ends = np.where(kind == 2, S + 1 + rng.integers(0, 50, B), ends).astype(np.int32)             # beyond the buffer
    starts = np.where(kind == 3, 0, starts).astype(np.int32)                                      # would read before the buffer
    ends = np.where(kind == 4, starts - 1 - rng.integers(0, 50, B), ends).astype(np.int32)        # reversed
    mid = rng.integers(1800, 2300, B).astype(np.uint32)
    n, mf, st = eng.mfcc_status(pcm, starts, ends, mid)
    n_bad = 0
    for b in range(B):
        oob = starts[b] < 1 or ends[b] > S or ends[b] < starts[b]
        short = not oob and ends[b] - starts[b] < fl
        if oob or short:
            assert st[b] == (ol.ST_SEG_OOB if oob else ol.ST_MFCC_FAIL) and n[b] == 0 and not mf[b].any(), (which, b, kind[b])
            n_bad += 1
            continue
        nn, m = orc.mfcc(pcm[b], int(starts[b]), int(ends[b]), ol.Atap(int(mid[b]), 0, 0, 0))
        if nn == 0:
            assert st[b] == ol.ST_MFCC_FAIL and n[b] == 0 and not mf[b].any(), (which, b)
            n_bad += 1
        else:
            assert st[b] == 0 and n[b] == nn and np.array_equal(mf[b, :nn], m) and not mf[b, nn:].any(), (which, b, int(starts[b]))
    assert 10 < n_bad < B - 30
    eng.close()


def test_extension_front_end_matches_its_oracle():
    """EXTENSION (no reference counterpart): BASELINE configs[4] front end -- 16 kHz, 320/160 framing, 512-point
    transform (2 x ST-style 256-point radix-4 + one radix-2 pass, oracle/q15_fft.c), 40 Mel, 12 MFCC.
    Whole path (VAD, MFCC, greedy DTW, argmin) against the parametrised oracle."""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(16)
    T, B, K, maxf = 100, 96, 12, 160
    cfg = dict(fs=16000, nfft=512, n_mel=40)
    orc = ol.Oracle(max_frames=maxf, **cfg)
    bank = synth.word_bank(8)
    tfr = [int(v) for v in rng.integers(70, 150, K)]
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(K) % 8, tfr, seed=21, bank=bank, rate=2,
                                                  S=synth.buf_len_for(max(tfr), 2)))
    tm = np.zeros((K, maxf + 1, 12), np.int16)
    for k in range(K):
        rc, a = orc.noise_atap(tp[k])
        seg = orc.vad(tp[k], a)
        n, m = orc.mfcc(tp[k], seg[0], seg[1], a)
        assert n == tfr[k]
        tm[k, :n] = m
    tf = np.array(tfr, np.uint32)
    pcm = synth.as_u16_numpy(synth.make_utterances(rng.integers(0, 8, B), [T] * B, seed=22, bank=bank, rate=2))
    pcm[:6] = synth.as_u16_numpy(synth.make_utterances(rng.integers(0, 8, 6), [T - 10] * 6, seed=23, bank=bank, rate=2,
                                                       S=pcm.shape[1], quiet_sigma=8.0, gain=3.0))
    pcm[6] = 2048
    eng = Engine(max_frames=maxf, device=0, **cfg)
    eng.set_templates_dense(tm, tf)
    out = eng.recognize(pcm)
    tpl = orc.make_templates(tm, tf)
    ores, omf, osc = orc.recognize_batch(pcm, tpl, n_threads=8)
    for b in range(B):
        rc, a = orc.noise_atap(pcm[b])
        seg = orc.vad(pcm[b], a)
        v = out["vad"][b]
        assert (v["mid_val"], v["n_thl"], v["z_thl"], v["s_thl"]) == a.astuple(), b
        assert np.array_equal(v["seg"], seg), b
    assert np.array_equal(out["mfcc"], omf)
    assert np.array_equal(out["scores"], osc)
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(out["results"][f], ores[f]), f
    assert (ores["frm_num"][7:] == T).all() and ores["status"][6] == ol.ST_VAD_FAIL
    eng.close()


@pytest.mark.parametrize("seed,maxf", [(1, 150), (2, 64), (3, 333)])
def test_extension_front_end_ragged_lengths(seed, maxf):
    """EXTENSION front end on ragged batches: frame counts from 1 to beyond the cap (k_mfcc_ext works on four frames at a
    time and reads the next batch ahead: partial groups, waves without frames, runs of empty work items, frame caps that
    are not a multiple of the 32-frame tile), silent captures in between; MFCC, scores and argmin against the oracle."""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(700 + seed)
    cfg = dict(fs=16000, nfft=512, n_mel=40)
    K, B = 9, 90
    orc = ol.Oracle(max_frames=maxf, **cfg)
    bank = synth.word_bank(6)
    tmax = min(maxf, 140)
    tfr = [int(v) for v in rng.integers(max(8, tmax // 2), tmax + 1, K)]
    S = synth.buf_len_for(int(tmax * 1.25) + 4, 2)
    tp = synth.as_u16_numpy(synth.make_utterances(np.arange(K) % 6, tfr, seed=31 + seed, bank=bank, rate=2, S=S))
    tm = np.zeros((K, maxf + 1, 12), np.int16)
    for k in range(K):
        rc, a = orc.noise_atap(tp[k])
        seg = orc.vad(tp[k], a)
        n, m = orc.mfcc(tp[k], seg[0], seg[1], a)
        assert n == tfr[k]
        tm[k, :n] = m
    tf = np.array(tfr, np.uint32)
    frames = [int(v) for v in rng.integers(8, int(tmax * 1.25) + 1, B)]          # some beyond the cap -> MFCC fail
    frames[:12] = [8, 9, 10, 11, 12, 13, 31, 32, 33, 63, 64, 65]
    pcm = synth.as_u16_numpy(synth.make_utterances(rng.integers(0, 6, B), frames, seed=41 + seed, bank=bank, rate=2, S=S))
    for b in (14, 15, 16, 40):
        pcm[b] = 2048 + b % 2                                                      # runs of captures without frames
    eng = Engine(max_frames=maxf, device=0, **cfg)
    eng.set_templates_dense(tm, tf)
    out = eng.recognize(pcm)
    tpl = orc.make_templates(tm, tf)
    ores, omf, osc = orc.recognize_batch(pcm, tpl, n_threads=8)
    assert np.array_equal(out["results"]["status"], ores["status"]) and np.array_equal(out["results"]["frm_num"], ores["frm_num"])
    assert len(set(ores["frm_num"].tolist())) > 20 and ((ores["status"] == ol.ST_MFCC_FAIL).any() or int(tmax * 1.25) <= maxf)
    assert np.array_equal(out["mfcc"], omf)
    assert np.array_equal(out["scores"], osc)
    for f in ("best_tpl", "min_dis"):
        assert np.array_equal(out["results"][f], ores[f]), f
    eng.close()


def test_dtw_generic_fallback_matches_oracle():
    """sequences too long for the LDS-staged kernel (max_frames = 6000) and more than 1024 templates go through
    the generic global-memory walk k_dtw: same arithmetic, must give the same scores"""
    from stm32_speech_recognition_amd import Engine
    rng = np.random.default_rng(5)
    # (a) huge frame cap, realistic lengths
    maxf, K, B = 6000, 7, 9
    orc = ol.Oracle(max_frames=maxf)
    tf = np.array([50, 900, 1500, 2999, 3000, 5999, 6000], np.uint32)
    tm = rng.integers(-2000, 2000, (K, maxf + 1, 12)).astype(np.int16)
    inf = np.array([60, 1000, 1499, 3000, 4000, 6000, 1, 2, 2500], np.uint32)
    im = rng.integers(-2000, 2000, (B, maxf, 12)).astype(np.int16)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf)
    sc, res = _dtw_all_modes(eng, im, inf)
    pad = np.zeros((1, 12), np.int16)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) for k in range(K)] for b in range(B)],
                    dtype=np.uint32)
    assert np.array_equal(sc, want)
    assert (want != ol.DIS_ERR).sum() >= 15
    eng.close()
    # (b) more templates than one workgroup can hold
    maxf, K, B = 40, 1100, 3
    orc = ol.Oracle(max_frames=maxf)
    tf = rng.integers(10, 41, K).astype(np.uint32)
    tm = rng.integers(-2000, 2000, (K, maxf + 1, 12)).astype(np.int16)
    inf = np.array([20, 30, 40], np.uint32)
    im = rng.integers(-2000, 2000, (B, maxf, 12)).astype(np.int16)
    eng = Engine(max_frames=maxf, device=0)
    eng.set_templates_dense(tm, tf)
    sc, res = _dtw_all_modes(eng, im, inf)
    want = np.array([[orc.dtw(np.concatenate([im[b], pad]), inf[b], tm[k], tf[k]) for k in range(K)] for b in range(B)],
                    dtype=np.uint32)
    assert np.array_equal(sc, want)
    assert np.array_equal(res["min_dis"], want.min(1))
    eng.close()


def test_argument_checks_and_edge_sizes(golden):
    """error behaviour of the batched API: no templates, bad alignment / stride, short buffers, empty batch;
    and the largest capture buffer the reference's u16 length allows (65 528 samples)"""
    import ctypes as C
    from stm32_speech_recognition_amd import Engine, SrError
    from stm32_speech_recognition_amd.engine import _vp
    e = Engine(max_frames=119, device=0)
    pcm = golden["pcm"]
    with pytest.raises(SrError, match="no templates"):
        e.recognize(pcm)
    e.set_templates_store(golden["store"])
    with pytest.raises(SrError, match="noise head|shorter"):
        e.recognize(pcm[:, :2000])
    assert e.recognize(pcm[:0])["results"].shape == (0,)  # empty batch is a no-op
    # device API: misaligned pointer / stride not a multiple of 8
    t = torch.zeros(2 * 16008 + 8, dtype=torch.int16, device="cuda:0")
    res = torch.zeros(2, 4, dtype=torch.int32, device="cuda:0")
    rc = e.L.sr_recognize_batch_dev(e.h, C.c_void_p(t.data_ptr() + 2), C.c_uint64(16008), C.c_uint32(16000), C.c_uint32(2),
                                    _vp(res), None, None, None, None)
    assert rc == 3
    rc = e.L.sr_recognize_batch_dev(e.h, C.c_void_p(t.data_ptr()), C.c_uint64(16004), C.c_uint32(16000), C.c_uint32(2),
                                    _vp(res), None, None, None, None)
    assert rc == 3
    # ragged: buf_len shorter than the row stride is fine
    out = e.recognize(np.concatenate([pcm, np.zeros((pcm.shape[0], 40), np.uint16)], 1), buf_len=16000)
    assert np.array_equal(out["results"]["min_dis"], golden["recg_dis"])
    # round-4 entry points: packed rows shorter than their payload, unknown development hook, DP variant out of range
    from stm32_speech_recognition_amd.engine import dev_hook, pack12
    pk = pack12(pcm)
    resb = np.zeros(len(pcm), dtype=[("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("d", "<u4")])
    rc = e.L.sr_recognize_batch_packed12(e.h, _vp(pk), C.c_uint64(pk.shape[1] - 1), C.c_uint32(pcm.shape[1]), C.c_uint32(len(pcm)),
                                         _vp(resb), None, None, None)
    assert rc == 3
    with pytest.raises(SrError, match="unknown development hook"):
        dev_hook("no_such_hook", 1)
    # ... and the PRODUCT library has no hooks at all: every name is refused
    assert e.L.sr_testing_build() == 0 and e.L.sr_dev_hook(b"cells_literal", C.c_int64(1)) == 3
    assert b"not compiled into the product library" in e.L.sr_last_error()
    with pytest.raises(SrError, match="lanes per pair"):
        e.set_dp_lanes(3)
    # a forced DTW geometry (development hooks: 16 utterances per workgroup, the whole tie table) gives the same results
    dev_hook("dtw_u", 16)
    dev_hook("dtw_tie_g", 32768)
    try:
        et = Engine(max_frames=119, device=0, testing=True)
        et.set_templates_store(golden["store"])
        out2 = et.recognize(pcm)
        et.close()
    finally:
        dev_hook("dtw_u", 0)
        dev_hook("dtw_tie_g", 0)
    assert np.array_equal(out2["results"]["min_dis"], golden["recg_dis"])
    e.close()
    # the 12-coefficient-only entry points say so on a generic front end with another feature width
    e13 = Engine(max_frames=40, device=0, n_mel=26, n_coef=13)
    e13.set_templates_dense(np.zeros((2, 41, 13), np.int16), np.array([20, 30], np.uint32))
    with pytest.raises(SrError, match="12-coefficient"):
        e13.dtw_dp(np.zeros((1, 40, 13), np.int16), np.array([20], np.uint32))
    e13.close()
    # maximum buffer: VAD(const u16 *vc, u16 buf_len, ...) -> 65 535 samples at most; 65 528 keeps rows 16-byte multiples
    o = ol.Oracle(max_frames=119)
    big = np.full((2, 65528), 2048, np.uint16)
    rng = np.random.default_rng(1)
    big[:] = np.clip(2048 + rng.normal(0, 5, big.shape), 0, 4095).astype(np.uint16)
    big[:, :16000] = pcm[:2]
    big[1, 60000:63000] = pcm[3, 3000:6000]  # a late second burst
    e2 = Engine(max_frames=119, device=0)
    vd = e2.vad(big)
    for b in range(2):
        rc_, a = o.noise_atap(big[b])
        assert np.array_equal(vd["seg"][b], o.vad(big[b], a))
    e2.close()


def test_development_hooks_change_geometry_not_results(golden):
    """sr_dev_hook (development knobs, read when an engine is created / a store is set): a forced frame-kernel grid, forced
    DTW workgroup shapes (utterances x templates per workgroup, tie-table size) -- same results as the golden fixture"""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import dev_hook
    pcm = golden["pcm"]
    for hooks in (dict(mfcc_grid=3), dict(dtw_u=1, dtw_kc=7), dict(dtw_u=9, dtw_tie_g=4096), dict(dtw_kc=1, dtw_debug=1)):
        for k, v in hooks.items():
            dev_hook(k, v)
        try:
            e = Engine(max_frames=119, device=0, testing=True)
            e.set_templates_store(golden["store"])
            out = e.recognize(np.tile(pcm, (9, 1)))
            e.close()
        finally:
            for k in hooks:
                dev_hook(k, 0)
        assert np.array_equal(out["results"]["min_dis"], np.tile(golden["recg_dis"], 9)), hooks
        assert np.array_equal(out["scores"][:len(pcm)], golden["recg_scores"]), hooks


def test_mel_term_fused_form_exhaustive(eng119):
    """k_mfcc's filterbank term -- ONE v_mul_hi_u32 of E << 4 with ceil(tri * 2^28 / 100) while every energy of the frame is
    <= floor(2^28 / 100) -- against the reference's u32 expression frq_spct[i]*tri[i]/(tri_top/10) (MFCC.C:139-161), swept on
    the device over EVERY weight 0..1000 (tri_top; the tables hold a subset) x EVERY energy 0..2 684 354, plus the weight
    recovered from the multiplier for the literal form.  Control: one past the certified range the fused form does fail."""
    import ctypes as C
    from stm32_speech_recognition_amd.engine import _vp
    emax = (1 << 28) // 100
    bad = np.zeros(1001, np.uint64)
    assert eng119.L.sr_mel_term_sweep(eng119.h, C.c_uint32(0), C.c_uint32(1001), C.c_uint32(emax), _vp(bad)) == 0
    assert not bad.any(), (np.nonzero(bad)[0][:8], bad[np.nonzero(bad)[0][:8]])
    # control: the bound is not slack by orders of magnitude -- some weight fails somewhere below 64 x the bound
    ctl = np.zeros(1001, np.uint64)
    assert eng119.L.sr_mel_term_sweep(eng119.h, C.c_uint32(0), C.c_uint32(1001), C.c_uint32(min(64 * emax, (1 << 28) - 1)), _vp(ctl)) == 0
    assert ctl.any()


def test_magnitude_cheap_form_exhaustive(eng119):
    """quiet frames of k_mfcc (every re^2 + im^2 <= 26 843) take (u32)(v_sqrt_f32(n) * 10) instead of the exactly corrected root:
    equal to the exact form -- itself certified against the host's sqrtf over all 2^32 inputs -- for EVERY n up to 65 535 on
    this device (the first difference is at n = 70 172); control: beyond that the cheap form does differ"""
    import ctypes as C
    from stm32_speech_recognition_amd.engine import _vp
    out = np.zeros(2, np.uint64)
    assert eng119.L.sr_mag_fast_sweep(eng119.h, C.c_uint32(65535), _vp(out)) == 0
    assert out[0] == 0 and out[1] == 0xFFFFFFFF, out
    assert eng119.L.sr_mag_fast_sweep(eng119.h, C.c_uint32((1 << 24) - 1), _vp(out)) == 0
    assert out[0] > 0 and out[1] > 26843 * 2, out


def test_magnitude_sweep_at_create_and_its_fallback():
    """sr_create sweeps the cheap magnitude form over its whole range on the device the engine runs on and falls back to the
    exactly corrected root for every frame if the chip's v_sqrt_f32 does not reproduce it (parity must not rest on
    a property of one stepping that only the test suite checks).  Here: the product reports the bound 70 171 on this device;
    an engine of the -DSR_TESTING build with the hook "mag_cheap_off" -- which makes sr_create behave as if the sweep had
    failed -- reports 0 and still gives the same MFCC bytes on captures that cover all three tiers."""
    from stm32_speech_recognition_amd import Engine
    from stm32_speech_recognition_amd.engine import dev_hook
    T, B = 64, 36
    bank = synth.word_bank(8)
    gains = np.repeat([0.6, 1.0, 2.4, 4.0], B // 4)
    pcm = np.concatenate([synth.as_u16_numpy(synth.make_utterances([i % 8], [T], seed=70 + i, bank=bank, gain=float(g)))
                          for i, g in enumerate(gains)])
    e = Engine(max_frames=T + 8, device=0)
    assert e.mag_cheap_bound() == 70171
    dev_hook("mag_cheap_off", 1)
    try:
        et = Engine(max_frames=T + 8, device=0, testing=True)
    finally:
        dev_hook("mag_cheap_off", 0)
    assert et.mag_cheap_bound() == 0
    # MFCC rows through the stage-level call of both engines
    v0 = e.vad(pcm)
    v1 = et.vad(pcm)
    assert np.array_equal(v0["seg"], v1["seg"]) and (v0["status"] == 0).all()
    n0, f0 = e.mfcc(pcm, v0["seg"][:, 0], v0["seg"][:, 1], v0["mid_val"])
    n1, f1 = et.mfcc(pcm, v1["seg"][:, 0], v1["seg"][:, 1], v1["mid_val"])
    assert (n0 == T).all() and np.array_equal(n0, n1) and np.array_equal(f0, f1)
    orc = ol.Oracle(max_frames=T + 8)
    tiers = orc.frame_tiers(pcm)
    assert min(tiers["quiet"], tiers["mid"], tiers["loud"]) > 0.05, tiers
    for b in (0, B // 2, B - 1):
        rc, a = orc.noise_atap(pcm[b])
        seg = orc.vad(pcm[b], a)
        n, m = orc.mfcc(pcm[b], seg[0], seg[1], a)
        assert np.array_equal(f0[b, :n], m)
    e.close()
    et.close()


def test_filterbank_fused_and_literal_forms_match_oracle():
    """captures at gains that put frames on both sides of the fused form's bound (and far past it, into the u32 wrap of the
    reference's product): MFCC rows identical to the oracle's, and the oracle's own spectra confirm both forms were taken"""
    from stm32_speech_recognition_amd import Engine
    T, B = 64, 48
    bank = synth.word_bank(8)
    gains = np.repeat([0.4, 1.0, 1.6, 2.5, 4.0, 7.0], B // 6)
    pcm = np.concatenate([synth.as_u16_numpy(synth.make_utterances(np.arange(8) % 8, [T] * 8, seed=50 + i, bank=bank, gain=float(g)))
                          for i, g in enumerate([0.4, 1.0, 1.6, 2.5, 4.0, 7.0])])
    assert pcm.shape[0] == B and len(gains) == B
    orc = ol.Oracle(max_frames=T + 8)
    eng = Engine(max_frames=T + 8, device=0)
    hamm = orc.tables()["hamm"].astype(np.int64)
    n_fast = n_slow = 0
    for mode in (1, 2):  # the 64-frame batch form and the 4-frame form of the frame kernel
        eng.set_small_launch(mode)
        vd = eng.vad(pcm)
        for b in range(B):
            rc, a = orc.noise_atap(pcm[b])
            seg = orc.vad(pcm[b], a)
            assert (vd["seg"][b][:2] == seg[:2]).all()
            if seg[1] < 0:
                continue
            n, m = orc.mfcc(pcm[b], seg[0], seg[1], a)
            gn, gm = eng.mfcc(pcm[b:b + 1], [seg[0]], [seg[1]], [a.mid_val])
            assert gn[0] == n and np.array_equal(gm[0, :n], m), (mode, b)
            if mode == 1:  # which form each frame takes, from the oracle's own spectrum
                x = pcm[b].astype(np.int64) - int(a.mid_val)
                for f in range(0, n, 7):
                    s0 = seg[0] + 80 * f
                    t = x[s0:s0 + 160] - np.trunc(x[s0 - 1:s0 + 159] * 95 / 100).astype(np.int64)
                    w = np.trunc(t * hamm / 1000).astype(np.int64).astype(np.int16)
                    mg = orc.fft_mag(w).astype(np.uint64)
                    if int(((mg * mg) & 0xFFFFFFFF).max()) <= (1 << 28) // 100:
                        n_fast += 1
                    else:
                        n_slow += 1
    eng.set_small_launch(0)
    eng.close()
    assert n_fast > 20 and n_slow > 20, (n_fast, n_slow)


def test_log_and_sqrt_device_functions_swept_directly(eng119):
    """(u32)(log(x)*100), (u32)sqrtf(x) and (u32)(sqrtf(r)*10) as the kernels compute them, against the same C
    expressions on the host: every step position of the log table +-1, perfect squares +-1 over the whole u32
    range (incl. the float-rounding regime >= 2^24), powers of two, and 2 M random values"""
    import ctypes as C
    from stm32_speech_recognition_amd.engine import _vp
    orc = ol.Oracle()
    rng = np.random.default_rng(99)
    k = np.arange(0, 65536, dtype=np.uint64)
    sq = (k * k).astype(np.uint64)
    cand = [np.array([0, 1, 2, 3, 0xFFFFFFFF, 0xFFFFFFFE, 0x7FFFFFFF, 0x80000000], np.uint64), sq, sq + 1, sq[1:] - 1,
            (np.uint64(1) << np.arange(32, dtype=np.uint64)), (np.uint64(1) << np.arange(1, 32, dtype=np.uint64)) - 1,
            rng.integers(0, 1 << 32, 1_000_000, dtype=np.uint64), rng.integers(0, 1 << 24, 500_000, dtype=np.uint64),
            (rng.integers(0, 65536, 500_000, dtype=np.uint64) ** 2 + rng.integers(-300, 300, 500_000)).clip(0, 0xFFFFFFFF)]
    # step positions of floor(100 ln n): n = ceil(exp(m/100)) and neighbours
    m = np.arange(0, 2219)
    th = np.ceil(np.exp(m / 100.0)).astype(np.uint64).clip(1, 0xFFFFFFFF)
    cand += [th, (th - 1).clip(0, None), (th + 1).clip(None, 0xFFFFFFFF)]
    x = np.concatenate([c.astype(np.uint64) for c in cand]).clip(0, 0xFFFFFFFF).astype(np.uint32)
    got = np.zeros(3 * len(x), np.uint32)
    want = np.zeros(3 * len(x), np.uint32)
    assert eng119.L.sr_math_diag(eng119.h, _vp(x), _vp(got), C.c_uint32(len(x))) == 0
    orc.L.sr_oracle_math_diag(x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_uint32(len(x)))
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (x[bad[:5] // 3], bad[:5] % 3, got[bad[:5]], want[bad[:5]])


def test_c_multi_gpu_demo(golden, tmp_path):
    """examples/multi_gpu_demo.c: sr_multi_* from a plain-C process on every GPU of the box (1 on the test box)"""
    import subprocess
    from test_abi_symbols import build_c_demo
    exe = str(tmp_path / "multi_gpu_demo")
    build_c_demo(exe, "multi_gpu_demo.c")
    golden["store"].tofile(str(tmp_path / "store.bin"))
    nb = len(golden["recg_best"])
    golden["pcm"][:nb].tofile(str(tmp_path / "caps.bin"))
    n = torch.cuda.device_count()
    out = subprocess.run([exe, str(tmp_path / "store.bin"), str(tmp_path / "caps.bin"), str(n)], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and f"on {n} device(s), gathered scores identical" in out.stdout, out.stdout + out.stderr
    for b in range(nb):
        assert f"capture {b}: slot {golden['recg_best'][b]} dis {golden['recg_dis'][b]} " in out.stdout


def test_reference_header_caller_links_the_library_and_prints_golden(golden, tmp_path):
    """tests/ref_caller/ref_caller.c includes the REFERENCE's OWN VAD.H / MFCC.H / DTW.H / ADC.H (not sr_engine.h), carries
    the main.c:258-295 sequence over an array store and is linked with -lsr_engine in place of the reference's objects
    (oracle/Makefile, built where /root/reference exists; the binary travels with the snapshot).  Run on the golden
    captures it must print what the reference's own objects produced: slot, distance, frame count, segment, thresholds
    and the first MFCC values."""
    import subprocess
    exe = os.path.join(os.path.dirname(ol.REF_PATH), "ref_caller")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/ref_caller is missing: build it where /root/reference exists (make -C oracle)")
    golden["store"].tofile(str(tmp_path / "store.bin"))
    caps = []
    nb = len(golden["recg_best"])
    for b in range(nb):
        golden["pcm"][b].tofile(str(tmp_path / f"cap{b}.bin"))
        caps.append(str(tmp_path / f"cap{b}.bin"))
    np.full(16000, 2048, np.uint16).tofile(str(tmp_path / "silence.bin"))
    out = subprocess.run([exe, str(tmp_path / "store.bin")] + caps + [str(tmp_path / "silence.bin")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sizeof(v_ftr_tag)=2860 vv_frm_max=119 VcBuf_Len=16000 atap_len=2400; store: 20 slots" in out.stdout
    lines = {ln.split(":")[0]: ln for ln in out.stdout.splitlines()[1:]}
    n_ok = 0
    for b in range(nb):
        ln = lines[caps[b]]
        if golden["recg_status"][b] != 0:
            assert "fail slot=-1 dis=4294967295" in ln
            continue
        n_ok += 1
        a, sg, m = golden["atap"][b], golden["seg"][b], golden["mfcc"][b][0]
        want = (f"slot={golden['recg_best'][b]} dis={golden['recg_dis'][b]} frm_num={golden['frm_num'][b]} seg=[{sg[0]},{sg[1]}) "
                f"mid={a[0]} n_thl={a[1]} z_thl={a[2]} s_thl={a[3]} mfcc0={m[0]},{m[1]},{m[2]}")
        assert want in ln, (want, ln)
    assert n_ok >= 8 and "VAD fail slot=-1 dis=4294967295" in lines[str(tmp_path / "silence.bin")]
    # the binary really is linked against the product library, not against the oracle's objects
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libsr_engine.so" in ldd and "libsr_ref" not in ldd and "liboracle" not in ldd


def test_c_demo_reference_call_pattern(golden, tmp_path):
    """plain-C program using the reference's own call sequence (noise_atap, VAD, get_mfcc, dtw per slot), the
    spch_recg drop-in and the batched API: all three must agree with each other and with the golden fixture"""
    import subprocess
    from test_abi_symbols import build_c_demo
    exe = str(tmp_path / "spch_recg_demo")
    build_c_demo(exe)
    golden["store"].tofile(str(tmp_path / "store.bin"))
    for b in (0, 3, 9):
        golden["pcm"][b].tofile(str(tmp_path / "cap.bin"))
        out = subprocess.run([exe, str(tmp_path / "store.bin"), str(tmp_path / "cap.bin")], capture_output=True, text=True,
                             timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert f"slot={golden['recg_best'][b]} dis={golden['recg_dis'][b]} " in out.stdout, out.stdout
        assert "us per call" in out.stdout and "us per capture" in out.stdout  # the demo's timing loops ran (same distances every time)
        print("\n".join(out.stdout.strip().splitlines()[-2:]))
    np.full(16000, 2048, np.uint16).tofile(str(tmp_path / "cap.bin"))  # silence: NULL + dis_err everywhere
    out = subprocess.run([exe, str(tmp_path / "store.bin"), str(tmp_path / "cap.bin")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "slot=-1 dis=4294967295" in out.stdout and "(null)" in out.stdout, out.stdout
