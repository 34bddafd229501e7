"""world_size-2 and -8 CPU tests (gloo) of the N>1 bookkeeping bench.py uses: contiguous utterance shards,
one all-gather of the [B_local, K] score matrix, max-over-ranks timing, first-minimum argmin."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from stm32_speech_recognition_amd import dist_util as du


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_scores(lo, hi, K):
    """deterministic u32 'DTW scores' of global utterances lo..hi-1, with ties and dis_err rows"""
    g = np.arange(lo, hi, dtype=np.uint64)[:, None]
    k = np.arange(K, dtype=np.uint64)[None, :]
    s = ((g * 2654435761 + k * 40503) % 977).astype(np.uint32)
    s[g[:, 0] % 7 == 3] = 0xFFFFFFFF          # utterances with no valid match
    s[:, 5] = 0xFFFFFFFF                       # an erased template slot
    return s


def _worker(rank, world, port, B_total, K, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    du.init_process_group("gloo")
    lo, hi = du.shard_bounds(B_total, world, rank)
    local = torch.from_numpy(_fake_scores(lo, hi, K).view(np.int32))
    gathered = du.all_gather_scores(local, world)
    t = du.max_over_ranks(1.0 + rank, "cpu", world)
    per_rank = du.gather_floats(10.0 + rank, "cpu", world)  # bench.py's per-rank step times
    assert per_rank == [10.0 + r for r in range(world)]
    best, mn = du.argmin_first(gathered)
    q.put((rank, lo, hi, gathered.numpy().view(np.uint32).copy(), t, best.numpy(), mn.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _fake_records(lo, hi, K):
    """the 16-byte result records (best_tpl, min_dis, frm_num, status) the argmin kernel would write for _fake_scores"""
    s = _fake_scores(lo, hi, K)
    best, mn = du.argmin_first(torch.from_numpy(s.view(np.int32)))
    r = np.zeros((hi - lo, 4), np.uint32)
    r[:, 0], r[:, 1], r[:, 2] = best.numpy(), mn.numpy(), 256
    return r


def _pipeline_worker(rank, world, port, B_local, K, steps, q, kind="scores"):
    """the double-buffered exchange bench.py runs at N > 1: step i's gather is in flight while step i+1 computes
    (kind "scores": the u32 score matrix, bench.py's default; "results": the 16-byte result records, bench.py --exchange results)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    du.init_process_group("gloo")
    make = _fake_scores if kind == "scores" else _fake_records
    K, K_gen = (K, K) if kind == "scores" else (4, K)
    _fs = lambda lo, hi, _k: make(lo, hi, K_gen)
    local = [torch.empty(B_local, K, dtype=torch.int32) for _ in range(2)]
    x = du.ScoreExchange(world, [torch.empty(world * B_local, K, dtype=torch.int32) for _ in range(2)])
    seen = []
    for i in range(steps):
        j = i % 2
        x.reserve(j)
        if i >= 2:  # the buffer about to be overwritten holds step i-2's complete gather
            seen.append((i - 2, x.gathered[j].numpy().view(np.uint32).copy()))
        off = 1000 * i
        local[j].copy_(torch.from_numpy(_fs(off + rank * B_local, off + (rank + 1) * B_local, K).view(np.int32)))
        x.launch(j, local[j])
    x.drain()
    for i in range(max(0, steps - 2), steps):
        seen.append((i, x.gathered[i % 2].numpy().view(np.uint32).copy()))
    q.put((rank, seen))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kind", ["scores", "results"])
@pytest.mark.parametrize("world", [2, 8])
def test_pipelined_exchange(world, kind):
    """the metric is quoted at 1/2/4/8 GPUs: the double-buffered exchange at world size 8 as well as 2, gathering the score
    matrix (north_star) or the 16-byte result records (bench.py --exchange results, SURVEY.md 8(e) names both)"""
    B_local, K, steps = 16, 6, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, B_local, K, steps, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen in outs:
        assert sorted(i for i, _ in seen) == list(range(steps))
        for i, g in seen:
            if kind == "scores":
                assert np.array_equal(g, _fake_scores(1000 * i, 1000 * i + world * B_local, K)), (rank, i)
            else:  # every rank's records, in global utterance order: (best_tpl, min_dis, frm_num, status) per utterance
                want = np.concatenate([_fake_records(1000 * i + r * B_local, 1000 * i + (r + 1) * B_local, K) for r in range(world)])
                assert g.shape == (world * B_local, 4) and np.array_equal(g, want), (rank, i)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            b = [du.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 8])
def test_allgather_of_scores(world):
    B_total, K = 64, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B_total, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_scores(0, B_total, K)
    for rank, lo, hi, gathered, t, best, mn in outs:
        per = B_total // world
        assert (lo, hi) == (rank * per, rank * per + per)
        assert np.array_equal(gathered, want)          # rank-major gather == global utterance order
        assert t == float(world)                       # max over ranks of (1 + rank)
        ref_mn = want.min(1)
        ref_best = np.array([int(np.argmax(row == row.min())) if row.min() != 0xFFFFFFFF else 0 for row in want])
        assert np.array_equal(mn, ref_mn.astype(np.int64)) and np.array_equal(best, ref_best)


def test_bench_plain_command_starts_one_rank_per_gpu_and_fails_loudly_without_devices():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE must start its own ranks.  Without
    a GPU here every rank refuses (there is no CPU path): non-zero exit code, nothing on stdout, both ranks named."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    if torch.cuda.is_available():
        return  # the GPU suite runs the real thing (test_bench_plain_command_launches_its_own_ranks)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "rank 0: device 0 requested" in p.stderr or "rank 1: device 1 requested" in p.stderr
    assert "stopping the other ranks" in p.stderr


def test_rank_without_master_port_is_refused(monkeypatch):
    """no silent default port: two jobs on one node would collide on it"""
    import pytest
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        du.init_process_group("gloo")
