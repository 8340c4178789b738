"""Multi-GPU layer: replica data parallelism over independent line crops / pages (one process per GPU).

The reference has no distributed code at all; the path shards naturally because every line crop is an
independent sequence (recognition/__init__.py:848-859 already flattens all lines of all pages into one sorted list).
Design for xGMI / RCCL (backend "nccl" IS RCCL on ROCm; "gloo" in the CPU tests):
  * no collective on the data path: each rank runs its own slot scheduler over its shard;
  * shard = round-robin deal over the width-sorted order, so every rank gets the same length mix (the sort is the
    length bucketing);
  * ONE fixed-size all_gather per call for the outputs: padded records [lines_per_rank, max_tokens + 1] x {token i32,
    score f32, bbox 6 x i32} (the extra row carries index + length) -- latency-bound (~0.4 MB / rank), single hop over the
    7 direct links;
  * weights: rank 0 repacks, one bucketed broadcast at start-up (<= 256 MB buckets; one-time, link-bound).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch


# A world of ONE rank normally short-circuits every function below (no process group needed). `force_collectives(True)` (bench.py
# --force-dist, SURYA_AMD_FORCE_COLLECTIVES=1) makes an initialised 1-rank group run the real collectives instead -- communicator
# set-up, device buffers, all_gather_into_tensor, broadcast -- so the RCCL path executes on a 1-GPU box exactly as rank 0 of an
# N-GPU job would run it.
_FORCE = [False]


def force_collectives(on: bool = True):
    _FORCE[0] = bool(on)


def collectives_on(group=None) -> bool:
    """True when the functions below go through torch.distributed: more than one rank, or a forced 1-rank group."""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _FORCE[0] or os.environ.get("SURYA_AMD_FORCE_COLLECTIVES") == "1"


def world_info(group=None) -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def collective_device(model_device, group=None) -> torch.device:
    """Where collective buffers must live: the GPU for RCCL ("nccl"), host memory for gloo."""
    import torch.distributed as dist
    return torch.device(model_device) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def assert_same_inputs(fingerprint: Sequence[int], group=None, device="cpu"):
    """Sharding a call only makes sense when every rank was handed the SAME inputs (each then recognises its share and
    all get all results). A job where every rank OCRs its own pages must not come here: compare a fingerprint of the
    inputs across ranks first and raise on every rank, instead of hanging in a mis-sized all_gather or silently mixing
    lines of different pages."""
    import torch.distributed as dist
    _, world = world_info(group)
    if not collectives_on(group):
        return
    mine = torch.tensor(list(fingerprint), dtype=torch.int64, device=device)
    allf = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allf, mine, group=group)
    if any(not torch.equal(f, allf[0]) for f in allf):
        raise RuntimeError("surya_amd: sharded call, but the ranks were given different inputs "
                           f"(fingerprints {[f.tolist() for f in allf]}); disable sharding (SURYA_AMD_SHARD=0 / "
                           "predictor.shard_lines = False) when every rank processes its own pages")


def shard_indices(n: int, world: int, rank: int) -> List[int]:
    """Positions (in the width-sorted order) handled by `rank`: a round-robin deal."""
    return list(range(rank, n, world))


class PackedLines(Sequence):
    """Ragged per-line values (token ids or scores) as ONE dense array [n, T] + lengths [n]; behaves like the list of lists the
    reference's prediction_loop returns (`lines[i]` is a Python list of line i's values) but converts a line only when somebody
    asks for it. The gather below hands every rank ALL lines of the call: unpacking them eagerly into Python lists on every rank
    was 10 us per line inside the timed step (22 ms at 8 x 256 lines) for a consumer that reads each line once."""

    __slots__ = ("data", "lens")

    def __init__(self, data: np.ndarray, lens: np.ndarray):
        self.data, self.lens = data, np.asarray(lens, np.int64)

    @classmethod
    def from_lists(cls, rows: Sequence[Sequence], width: int, dtype) -> "PackedLines":
        n = len(rows)
        lens = np.fromiter((min(len(r), width) for r in rows), np.int64, n)
        data = np.zeros((n, width), dtype)
        if n and int(lens.sum()):
            mask = np.arange(width)[None, :] < lens[:, None]
            data[mask] = np.fromiter((x for r, L in zip(rows, lens) for x in r[:L]), dtype, int(lens.sum()))
        return cls(data, lens)

    def __len__(self) -> int:
        return len(self.lens)

    def row(self, i: int) -> np.ndarray:
        """Line i as an array view (no conversion)."""
        return self.data[i, : int(self.lens[i])]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.data[i, : int(self.lens[i])].tolist()

    def __iter__(self):
        for i in range(len(self)):
            yield self.data[i, : int(self.lens[i])].tolist()

    def __eq__(self, other):
        if isinstance(other, PackedLines):
            other = list(other)
        return list(self) == other

    def tolists(self) -> list:
        return list(self)


def gather_line_outputs(tokens, scores, bboxes: np.ndarray, local_idx: Sequence[int], n_total: int, max_tokens: int,
                        device="cpu", group=None, stats: dict | None = None):
    """All ranks contribute the outputs of their shard; every rank gets all n_total lines back in global order.
    tokens / scores: lists of per-line lists, or PackedLines (the continuous-batching loop's own dense bookkeeping,
    RecognitionPredictor.last_packed -- no per-token Python on the way in). bboxes: [n_local, max_tokens, 6].
    Returns (tokens PackedLines, scores PackedLines, bboxes [n_total, max_tokens, 6]). `stats` (optional dict) accumulates host seconds.
    ONE collective: a rank's record is a flat int32 buffer of four planes -- meta [per_rank, 2] (global index + 1, length; 0 =
    padding line), token ids [per_rank, T], score bits [per_rank, T], bbox ints [per_rank, T, 6]; entries past a line's length are
    zero. Planes, not an array of per-token structs: packing and unpacking are then contiguous whole-array copies (the [.., 8]
    struct layout of round 4 spent 5 ms of strided numpy writes at 2048 lines); nothing here is per line."""
    import time
    import torch.distributed as dist
    t_start = time.perf_counter()
    rank, world = world_info(group)
    per_rank = (n_total + world - 1) // world
    T = max_tokens
    o_tok, o_sc, o_bb, rec_len = 2 * per_rank, (2 + T) * per_rank, (2 + 2 * T) * per_rank, (2 + 8 * T) * per_rank
    rec = np.zeros(rec_len, np.int32)
    n_local = len(local_idx)
    if n_local:
        pt = tokens if isinstance(tokens, PackedLines) else PackedLines.from_lists(tokens, T, np.int32)
        ps = scores if isinstance(scores, PackedLines) else PackedLines.from_lists(scores, T, np.float32)
        assert len(pt) == n_local and len(ps) == n_local
        lens = np.minimum(pt.lens, T)
        w = min(T, pt.data.shape[1], ps.data.shape[1])
        meta = rec[:o_tok].reshape(per_rank, 2)
        meta[:n_local, 0] = np.asarray(local_idx, np.int64) + 1
        meta[:n_local, 1] = lens
        mask = np.arange(T)[None, :] < lens[:, None]                       # [n_local, T]
        tok_p = rec[o_tok:o_sc].reshape(per_rank, T)
        sc_p = rec[o_sc:o_bb].reshape(per_rank, T).view(np.float32)         # bit-cast, lossless
        bb_p = rec[o_bb:].reshape(per_rank, T, 6)
        np.multiply(pt.data[:, :w], mask[:, :w], out=tok_p[:n_local, :w], casting="unsafe")
        np.copyto(sc_p[:n_local, :w], ps.data[:, :w], casting="unsafe")
        sc_p[:n_local][~mask] = 0
        np.copyto(bb_p[:n_local], bboxes[:, :T], casting="unsafe")         # float box coordinates are whole numbers (bbox bins)
        bb_p[:n_local][~mask] = 0
    t_packed = time.perf_counter()
    if not collectives_on(group):
        allr = rec[None]
    else:
        mine = torch.from_numpy(rec).to(device)
        out = torch.empty((world * rec_len,), dtype=mine.dtype, device=mine.device)    # concatenated along dim 0
        dist.all_gather_into_tensor(out, mine, group=group)
        allr = out.cpu().numpy().reshape(world, rec_len)
    t_gathered = time.perf_counter()
    W = allr.shape[0]
    meta = allr[:, :o_tok].reshape(W * per_rank, 2)
    rows = np.flatnonzero(meta[:, 0] > 0)
    g = meta[rows, 0].astype(np.int64) - 1
    dense = len(rows) == W * per_rank                                      # no padding lines: plain permutations
    def plane(lo, hi, shape):
        a = allr[:, lo:hi].reshape((W * per_rank,) + shape)
        return a if dense else a[rows]
    out_tok = np.zeros((n_total, T), np.int32)
    out_sc = np.zeros((n_total, T), np.float32)
    out_bb = np.zeros((n_total, T, 6), np.float32)
    out_len = np.zeros(n_total, np.int64)
    out_tok[g] = plane(o_tok, o_sc, (T,))
    out_sc[g] = plane(o_sc, o_bb, (T,)).view(np.float32)
    out_bb[g] = plane(o_bb, rec_len, (T, 6))
    out_len[g] = meta[rows, 1]
    if stats is not None:                  # host seconds: pack + unpack (whole-array numpy) apart from the collective (H2D, all_gather, D2H)
        stats["pack_unpack_s"] = stats.get("pack_unpack_s", 0.0) + (t_packed - t_start) + (time.perf_counter() - t_gathered)
        stats["collective_s"] = stats.get("collective_s", 0.0) + (t_gathered - t_packed)
    return PackedLines(out_tok, out_len), PackedLines(out_sc, out_len), out_bb


def gather_objects(local: list, local_idx: Sequence[int], n_total: int, group=None) -> list:
    """Small picklable per-item results (detection boxes of a page): every rank gets all n_total back in global order."""
    import torch.distributed as dist
    _, world = world_info(group)
    if not collectives_on(group):
        return list(local)
    parts = [None] * world
    dist.all_gather_object(parts, (list(local_idx), list(local)), group=group)
    out = [None] * n_total
    for idx, items in parts:
        for i, it in zip(idx, items):
            out[i] = it
    return out


def broadcast_tensors(tensors: Sequence[torch.Tensor], src: int = 0, bucket_bytes: int = 256 << 20, group=None):
    """In-place broadcast of a list of same-device tensors in flat buckets (few large collectives: xGMI links are
    point-to-point, ~153 GB/s each, so bandwidth comes from message size, not from message count)."""
    import torch.distributed as dist
    _, world = world_info(group)
    if not collectives_on(group):
        return
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        bucket, size = [], 0
        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.cat([t.reshape(-1) for t in bucket])
            dist.broadcast(flat, src, group=group)
            off = 0
            for t in bucket:
                n = t.numel()
                t.copy_(flat[off: off + n].view_as(t))
                off += n
            bucket, size = [], 0
        for t in ts:
            nb = t.numel() * t.element_size()
            if size and size + nb > bucket_bytes:
                flush()
            bucket.append(t)
            size += nb
        flush()


def share_weights(weights, device, src: int = 0, group=None, bucket_bytes: int = 256 << 20) -> List[torch.Tensor]:
    """Weight distribution at start-up (north_star: "RCCL broadcast of weights"): rank `src` repacked the checkpoint into
    the kernel layout and passes its tensor list; every other rank passes None, learns shapes / dtypes from a broadcast
    manifest, allocates on `device` and receives the bytes through broadcast_tensors. Returns the list on every rank.
    One checkpoint read + one repack per node instead of one per GPU; REC-FULL is ~1.35 GB in 6 buckets."""
    import torch.distributed as dist
    rank, world = world_info(group)
    if not collectives_on(group):
        return list(weights)
    manifest = [[(tuple(t.shape), t.dtype) for t in weights] if rank == src else None]
    dist.broadcast_object_list(manifest, src=src, group=group)
    cdev = collective_device(device, group)
    if rank == src:
        local = [t.to(cdev) for t in weights]
    else:
        local = [torch.empty(shape, dtype=dt, device=cdev) for shape, dt in manifest[0]]
    broadcast_tensors(local, src=src, bucket_bytes=bucket_bytes, group=group)
    if rank == src:
        return list(weights)
    return [t.to(device) for t in local]
