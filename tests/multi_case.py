"""Shared body of the sr_multi_* tests (C ABI, one process driving several devices): sharded recognition + all-gather of
the score matrix == the single-engine answer, through the host-buffer call and the device-resident call.

Importable (tests/test_gpu_parity.py) and runnable:  python tests/multi_case.py 0,0,0 37
The script form exists for the fake-RCCL runs: csrc/sr_multi.cpp binds its collective library once per process, so a run
against tests/fake_rccl/librccl.so.1 (SR_RCCL_LIBRARY) with several ranks on device 0 (development hook
"multi_allow_dup", switched on here when the device list holds duplicates) needs a process of its own."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE_RCCL = os.path.join(HERE, "fake_rccl", "librccl.so.1")


def multi_case(devices, golden, B=37):
    import torch

    from stm32_speech_recognition_amd.engine import Engine, MultiEngine, results_from_torch
    n = len(devices)
    g = golden["pcm"]                                                   # 12 captures
    pcm = np.concatenate([g, g[::-1], g[3:], g[::-1][5:]])[:B]          # B = 37: uneven shards on 2+ devices
    assert len(pcm) == B
    e1 = Engine(device=devices[0])
    e1.set_templates_store(golden["store"])
    want = e1.recognize(pcm, want_mfcc=False, want_vad=False)
    # duplicate device ordinals (several "ranks" on one GPU over the in-process collective double) need the development
    # hook multi_allow_dup, which exists only in the -DSR_TESTING build of the library; the single engine the answer is
    # compared with stays on the product library
    me = MultiEngine(devices, testing=len(set(devices)) < len(devices))
    me.set_templates_store(golden["store"])
    res, sc = me.recognize(pcm)                                         # padded / empty last shards when n does not divide B
    for f in ("best_tpl", "min_dis", "frm_num", "status"):
        assert np.array_equal(res[f], want["results"][f]), f
    assert np.array_equal(sc, want["scores"])
    # device-resident shards, equal size: every device ends up with the whole score matrix
    Bp, K = min(12, B // n), me.K
    if Bp:
        pl, rl, al = [], [], []
        for i, d in enumerate(devices):
            dev = torch.device("cuda", d)
            pl.append(torch.from_numpy(pcm[i * Bp:(i + 1) * Bp].view(np.int16)).to(dev))
            rl.append(torch.zeros(Bp, 4, dtype=torch.int32, device=dev))
            al.append(torch.full((n * Bp, K), -1, dtype=torch.int32, device=dev))
        for _ in range(2):                                              # twice: the second gather overwrites a complete matrix
            me.recognize_dev(pl, rl, al)
        for i in range(n):
            assert np.array_equal(al[i].cpu().numpy().view(np.uint32), want["scores"][:n * Bp]), i
            r = results_from_torch(rl[i])
            assert np.array_equal(r["best_tpl"], want["results"]["best_tpl"][i * Bp:(i + 1) * Bp])
            assert np.array_equal(r["min_dis"], want["results"]["min_dis"][i * Bp:(i + 1) * Bp])
    # a template upload that fails on the first device keeps every device on the old store and the handle usable
    bad = np.zeros((3, 5, 12), np.int16)
    try:
        me.set_templates_dense(bad, np.array([4, 0x7FFFFFFF, 4], np.uint32))
        raise AssertionError("an impossible frame count was accepted")
    except Exception as e:  # noqa: BLE001
        assert "sr_multi error" in str(e), e
    me.K = K
    res2, sc2 = me.recognize(pcm)
    assert np.array_equal(sc2, want["scores"]) and np.array_equal(res2["best_tpl"], want["results"]["best_tpl"])
    me.close()
    e1.close()
    return dict(n=n, B=B, Bp=Bp, K=int(K))


def fake_stats():
    L = C.CDLL(FAKE_RCCL)
    out = (C.c_uint64 * 2)()
    L.fake_rccl_stats(out)
    return int(out[0]), int(out[1])


if __name__ == "__main__":
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    devs = [int(x) for x in sys.argv[1].split(",")]
    if len(set(devs)) < len(devs):
        from stm32_speech_recognition_amd.engine import dev_hook
        dev_hook("multi_allow_dup", 1)
    Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [37]
    g = np.load(os.path.join(HERE, "golden", "ref_golden.npz"))
    out = [multi_case(devs, g, B) for B in Bs]
    rep = dict(cases=out)
    if os.environ.get("SR_RCCL_LIBRARY"):
        rep["fake_allgathers"], rep["fake_copies"] = fake_stats()
    print("MULTI_CASE " + json.dumps(rep))
