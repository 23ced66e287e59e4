"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.

What is exercised is the sharding and the exchange -- `partition_by_bytes`, the count all-gather and the
grouped send / recv of the token buffers at their exact lengths (and the earlier padded gather; tiktoken_amd/distributed.py, the same functions bench.py uses with
the nccl/RCCL backend).  There is no CPU encode path in the product, so each rank's *encoder* here is the
C oracle (tests may use it); the assertion is that the gathered result equals the oracle's encoding of
the undivided batch, i.e. sharding + gather preserve order and content."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as h
from tiktoken_amd.distributed import encode_ordinary_batch_sharded, exchange_verdicts, gather_tokens, ids_digest, partition_by_bytes, verify_gathered


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C = h.c_oracle_for("cl100k_shaped")
        blob, off = h.gen_corpus(0xD157, 0, 1 << 19, threads=1)
        res = encode_ordinary_batch_sharded(lambda b, o: C.encode_batch(b, o, None, 1), blob, off, rank, world, dist, torch)
        if rank == 0:
            toks, toff = res
            rt, ro = C.encode_batch(blob, off, None, 1)
            q.put(bool(np.array_equal(toks, rt) and np.array_equal(toff, ro)))
        else:
            assert res is None
        # the pipelined form bench.py uses for N > 1: the rank's documents in sub-batches, each gathered asynchronously
        first, last = partition_by_bytes(off, world)[rank]
        mine_off = off[first:last + 1]
        pend = []
        for a, b in partition_by_bytes(mine_off - mine_off[0], 3):
            lo, hi = int(mine_off[a]), int(mine_off[b])
            toks, _ = C.encode_batch(blob[lo:hi], (mine_off[a:b + 1] - mine_off[a]).astype(np.uint64), None, 1)
            t = torch.from_numpy(np.ascontiguousarray(toks).view(np.int32).copy())
            pend.append(gather_tokens(t, len(toks), rank, world, dist, torch, async_op=True))
        got = [p.wait() for p in pend]
        if rank == 0:
            per_rank = [np.concatenate([g[0][r].numpy().view(np.uint32) for g in got]) for r in range(world)]
            rt, _ = C.encode_batch(blob, off, None, 1)
            q.put(bool(np.array_equal(np.concatenate(per_rank), rt)))
        else:
            assert all(g[0] is None for g in got)
        # what bench.py --gpus N does after its timed region: every rank checks its OWN shard against the oracle and digests its id
        # stream, the verdicts travel in one all-gather, and the destination compares what it RECEIVED per peer with what was sent
        lo, hi = int(mine_off[0]), int(mine_off[-1])
        # (the rank's "encoder" is the piece-by-piece Python restatement here, the checker the C one: two implementations, as on the GPU box)
        doc_lo = mine_off - mine_off[0]
        own = np.concatenate([np.asarray(h.py_oracle.encode_ordinary(blob[lo + int(a):lo + int(b)].tobytes().decode(), h.PAT_STR[1], h.load_vocab("cl100k_shaped")), np.uint32)
                              for a, b in zip(doc_lo[:-1], doc_lo[1:])] + [np.zeros(0, np.uint32)])
        ref, _ = C.encode_batch(blob[lo:hi], doc_lo.astype(np.uint64), None, 1)
        cnt, dig = ids_digest(own)
        verdicts = exchange_verdicts(cnt, dig, bool(np.array_equal(own, ref)), rank, world, dist, torch)
        assert verdicts[rank] == (cnt, dig, True) and all(v[2] for v in verdicts)
        # a rank whose shard is wrong says so, and every rank hears it
        wrong = own.copy()
        if rank == world - 1 and len(wrong):
            wrong[len(wrong) // 3] ^= 1
        bad = exchange_verdicts(*ids_digest(wrong), bool(np.array_equal(wrong, ref)), rank, world, dist, torch)
        assert [v[2] for v in bad] == [True] * (world - 1) + [False] and bad[-1][1] != verdicts[-1][1]
        parts, counts = gather_tokens(torch.from_numpy(np.ascontiguousarray(own).view(np.int32).copy()), len(own), rank, world, dist, torch)
        if rank == 0:
            good = verify_gathered(parts, verdicts)
            # ... and a gather that went wrong is seen: one peer's ids delivered twice, a flipped id, a dropped tail
            twice = verify_gathered([parts[0]] * world, verdicts)
            flipped = [p.clone() for p in parts]
            flipped[-1][len(flipped[-1]) // 2] ^= 1
            short = list(parts[:-1]) + [parts[-1][:-1]]
            q.put(bool(good["gather_verified"] and counts == [v[0] for v in verdicts] and not twice["gather_verified"]
                       and not verify_gathered(flipped, verdicts)["gather_verified"] and not verify_gathered(short, verdicts)["gather_verified"]))
        # a rank without tokens; the padded form
        for padded in (False, True):
            t = torch.arange(5 if rank == 0 else 0, dtype=torch.int32) + 100 * rank
            parts, counts = gather_tokens(t, t.numel(), rank, world, dist, torch, padded=padded)
            assert counts == [5] + [0] * (world - 1)
            t = torch.arange(3 + rank, dtype=torch.int32) + 100 * rank
            parts, counts = gather_tokens(t, t.numel(), rank, world, dist, torch, padded=padded)
            if rank == 0:
                q.put(bool(all(np.array_equal(parts[r].numpy(), np.arange(3 + r) + 100 * r) for r in range(world))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = all(q.get(timeout=150) for _ in range(5))  # plain and pipelined exchange; the digest exchange; exact-length and padded form with uneven counts
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok


def test_partition_by_bytes_is_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 5000, size=1000)
    off = np.zeros(1001, np.uint64)
    off[1:] = np.cumsum(lens)
    for world in (1, 2, 3, 8):
        parts = partition_by_bytes(off, world)
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(parts[:-1], parts[1:]))
        sizes = [int(off[b] - off[a]) for a, b in parts]
        assert max(sizes) - min(sizes) <= 2 * 5000
    assert partition_by_bytes(np.zeros(1, np.uint64), 4) == [(0, 0)] * 4  # empty batch
    assert partition_by_bytes(np.array([0, 10], np.uint64), 4)[-1] == (0, 1) or sum(b - a for a, b in partition_by_bytes(np.array([0, 10], np.uint64), 4)) == 1
