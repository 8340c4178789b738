"""CPU, world_size 2 over gloo: the N>1 path (sharding, output all_gather, weight broadcast) without GPUs.
A deterministic stand-in plays the per-rank recogniser; the contract checked is the one of SURVEY 8(e):
outputs of 2 ranks == outputs of 1 rank for any sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from surya_amd import dist as sd


def _fake_line(i, max_tokens):
    rng = np.random.default_rng(1000 + i)
    L = int(rng.integers(0, max_tokens + 1))                  # includes empty lines
    toks = rng.integers(0, 70000, size=L).tolist()
    scores = rng.random(L).astype(np.float32).tolist()
    bb = np.zeros((max_tokens, 6), np.float32)
    bb[:L] = rng.integers(0, 1025, size=(L, 6))
    return toks, scores, bb


NS = (7, 16, 1)       # odd split, even split, fewer lines than ranks


def _worker(rank, world, port, max_tokens, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for n in NS:
        idx = sd.shard_indices(n, world, rank)
        lines = [_fake_line(i, max_tokens) for i in idx]
        bbs = np.stack([l[2] for l in lines]) if lines else np.zeros((0, max_tokens, 6), np.float32)
        out.append(sd.gather_line_outputs([l[0] for l in lines], [l[1] for l in lines], bbs, idx, n, max_tokens))
    w = [torch.full((5, 3), float(rank + 1)), torch.arange(7, dtype=torch.int32) * (rank + 1), torch.full((2,), 9.0 * (rank + 1))]
    sd.broadcast_tensors(w, src=0, bucket_bytes=32)           # tiny buckets: exercises the bucketing
    q.put((rank, out, [t.tolist() for t in w]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    max_tokens = 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, max_tokens, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, w in results:
        for n, (tok, sc, bb) in zip(NS, out):
            ref = [_fake_line(i, max_tokens) for i in range(n)]
            assert tok == [r[0] for r in ref]
            for a, b in zip(sc, [r[1] for r in ref]):
                assert a == b                                  # fp32 scores travel bit-exactly
            assert np.array_equal(bb, np.stack([r[2] for r in ref]))
        assert w[0] == [[1.0] * 3] * 5 and w[1] == list(range(7)) and w[2] == [9.0, 9.0]     # rank 0's weights everywhere


def test_shard_is_a_partition():
    for n in (0, 1, 5, 256):
        for world in (1, 2, 4, 8):
            parts = [sd.shard_indices(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
