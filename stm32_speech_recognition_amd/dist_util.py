"""Multi-GPU plumbing: one process per GPU, utterances sharded over ranks, templates replicated,
ONE collective per step (all-gather of the per-template score matrix).  torch.distributed backend
"nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (tests/test_sharding_gloo.py)."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, local_rank=0):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # every rank must name the SAME rendezvous port, so a rank cannot probe for a free one by itself: the launcher
    # (torch.distributed.run, or bench.py's own self_launch, which probes 127.0.0.1 for a free port) hands it over.
    if "MASTER_PORT" not in os.environ:
        raise RuntimeError("MASTER_PORT is not set: start the ranks with torch.distributed.run or `python bench.py --gpus N` "
                           "(which picks a free port), or export one port for all ranks")
    # the host driver of these boxes only supports dmabuf IPC: without this RCCL / cross-process tensor sharing fails with
    # "hipIpcGetMemHandle: invalid argument" (stated for this image; already exported in its environment, kept as a default)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)


def shard_bounds(n_total, world, rank):
    """Contiguous utterance range [lo, hi) of `rank`; sizes differ by at most 1 (ragged totals allowed)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_scores(local_scores, world, out=None):
    """local_scores: [B_local, K] (same B_local on every rank) -> [world*B_local, K], rank-major,
    i.e. global utterance order when shards are contiguous.  One collective."""
    if world == 1:
        return local_scores
    if out is None:
        out = torch.empty((world * local_scores.shape[0],) + tuple(local_scores.shape[1:]), dtype=local_scores.dtype,
                          device=local_scores.device)
    dist.all_gather_into_tensor(out, local_scores.contiguous())
    return out


class ScoreExchange:
    """Double-buffered exchange step: the all-gather of step i's scores runs on the collective's own stream while
    the kernels of step i+1 fill the other buffer (the collective is ~1 ms of xGMI traffic next to ~24 ms of
    kernels, but serialising it would still cost a few percent of scaling efficiency).

        x = ScoreExchange(world, [gathered0, gathered1])
        per step i:  j = i % 2;  x.reserve(j);  <kernels write local_scores[j]>;  x.launch(j, local_scores[j])
        at the end:  x.drain()
    reserve(j) orders the current stream after the previous collective that read local_scores[j] / wrote
    gathered[j] (Work.wait() blocks the stream, not the host, on the nccl backend)."""

    def __init__(self, world, gathered, force=False):
        self.world = world
        self.force = force  # run the collective even at world == 1 (tests: the RCCL calls themselves on a 1-GPU box)
        self.gathered = list(gathered)
        self.pending = [None] * len(self.gathered)

    def reserve(self, j):
        if self.pending[j] is not None:
            self.pending[j].wait()
            self.pending[j] = None

    def launch(self, j, local_scores):
        if self.world == 1 and not self.force:
            return
        self.reserve(j)
        self.pending[j] = dist.all_gather_into_tensor(self.gathered[j], local_scores.contiguous(), async_op=True)

    def drain(self):
        for j in range(len(self.pending)):
            self.reserve(j)


def max_over_ranks(value, device, world):
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device, world):
    """one float per rank -> list of all ranks' values, in rank order (diagnostics: per-rank step times)"""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    n = dist.get_world_size()
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = torch.empty(n, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu()]


def argmin_first(scores):
    """Strict-< scan in slot order (main.c:285): first minimum wins, all dis_err -> slot 0.
    scores: int32 tensor holding u32 bit patterns."""
    u = scores.to(torch.int64) & 0xFFFFFFFF
    mn, idx = torch.min(u, dim=1)
    # torch.min returns an arbitrary index among equal minima on some backends: recompute the first
    first = (u == mn.unsqueeze(1)).to(torch.int64).argmax(dim=1)
    first = torch.where(mn == 0xFFFFFFFF, torch.zeros_like(first), first)
    return first, mn
