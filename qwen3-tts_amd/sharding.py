"""Multi-GPU partitioning of the vocoder path: independent units, no exchange step.

In the reference every ``max_tokens``-frame unit of every request is decoded with no carried state
(src/worker/vocoder.rs:65-82), so both a batch of utterances and the segments of one long stream
shard as contiguous ranges of units, one process per GPU.  The only communication is the optional
gather of finished waveform shards to rank 0: ``UnitGather`` (one ``torch.distributed.gather`` of device
buffers -- RCCL over xGMI with the ``nccl`` backend, CPU tensors with ``gloo`` in the tests), or
``gather_stream`` for host arrays (``gather_object``: pickled through the process group, not a device
collective).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

SAMPLES_PER_TOKEN = 1920


def unit_spans(n_frames: int, unit_frames: int) -> List[Tuple[int, int]]:
    """[(start_frame, end_frame)] of the reference's chunk loop (vocoder.rs:65-67)."""
    return [(s, min(s + unit_frames, n_frames)) for s in range(0, n_frames, unit_frames)]


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of n_items for this rank (first n_items % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def decode_stream_shard(synthesize: Callable[[np.ndarray], np.ndarray], codes: np.ndarray, unit_frames: int,
                        rank: int, world: int) -> np.ndarray:
    """This rank's part of one long code stream [n_frames,16]: the frames of its contiguous unit range,
    decoded with the ordinary chunking.  Because shards start on unit boundaries, concatenating the
    ranks' results in rank order is bit-identical to a single-GPU decode ("seam-exact stitching")."""
    spans = unit_spans(codes.shape[0], unit_frames)
    lo, hi = shard_range(len(spans), rank, world)
    if lo == hi:
        return np.zeros(0, dtype=np.float32)
    return synthesize(codes[spans[lo][0]: spans[hi - 1][1]])


def gather_stream(local: np.ndarray, rank: int, world: int, dist=None) -> np.ndarray | None:
    """Assemble the shards on rank 0 (returns None elsewhere).  dist = torch.distributed (initialised)."""
    if world == 1:
        return local
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0)
    if rank != 0:
        return None
    return np.concatenate(parts)


class UnitGather:
    """Gather of decoded units to rank 0 as ONE collective on tensors (device buffers with the ``nccl`` backend: RCCL over
    xGMI; CPU tensors with ``gloo``), not pickled objects: every rank contributes a [max_units, stride] buffer (its shard,
    zero-padded to the largest shard) and rank 0 receives world of them.  Buffers are allocated once, so a timed loop
    allocates nothing.  ``stitch`` cuts the kept samples of each unit (vocoder.rs:79-80) out of the gathered rows in rank
    order -- shards start on unit boundaries, so the result is bit-identical to a single-GPU decode."""

    def __init__(self, n_units_total: int, stride: int, rank: int, world: int, dist, device):
        import torch
        self.rank, self.world, self.dist = rank, world, dist
        self.ranges = [shard_range(n_units_total, r, world) for r in range(world)]
        self.max_units = max(hi - lo for lo, hi in self.ranges)
        self.pad = torch.zeros((self.max_units, stride), dtype=torch.float32, device=device)
        self.parts = [torch.empty_like(self.pad) for _ in range(world)] if rank == 0 else None

    def gather(self, local, n_local: int):
        """local: [>= n_local, stride] tensor holding this rank's decoded units (complete on its device)."""
        if n_local == self.max_units and local.is_contiguous() and local.shape[1] == self.pad.shape[1]:
            # a full-sized shard is already the buffer the collective wants: no staging copy (a rank with the largest
            # share -- every rank when the units divide evenly -- sends its decode output as it lies)
            self.dist.gather(local[:n_local], self.parts, dst=0)
            return
        self.pad[:n_local] = local[:n_local]
        self.dist.gather(self.pad, self.parts, dst=0)

    def stitch(self, keep: Sequence[int]) -> np.ndarray:
        """Rank 0: the stream's samples; keep[i] = samples kept of global unit i."""
        out = []
        for r, (lo, hi) in enumerate(self.ranges):
            g = self.parts[r].cpu().numpy()
            out.extend(g[i - lo, : keep[i]] for i in range(lo, hi))
        return np.concatenate(out) if out else np.zeros(0, dtype=np.float32)


def shard_utterances(n_utts: int, rank: int, world: int) -> range:
    lo, hi = shard_range(n_utts, rank, world)
    return range(lo, hi)


# ---- encoder path: clips are independent, so a batch shards as contiguous clip ranges -------------
def encode_clips_shard(encode_batch: Callable[[Sequence[np.ndarray]], List[np.ndarray]], clips: Sequence[np.ndarray],
                       rank: int, world: int) -> List[np.ndarray]:
    """This rank's part of a batch of clips: its contiguous range, encoded in one pass."""
    lo, hi = shard_range(len(clips), rank, world)
    return list(encode_batch(clips[lo:hi])) if hi > lo else []


def gather_codes(local: List[np.ndarray], rank: int, world: int, dist=None) -> List[np.ndarray] | None:
    """Assemble the per-clip code arrays ([n_tokens_i, 16] int64, ragged) on rank 0 in clip order (None elsewhere)."""
    if world == 1:
        return local
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0)
    if rank != 0:
        return None
    return [c for p in parts for c in p]
