"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The encode path shards by document and needs no data-path collective (documents never interact,
reference tiktoken/core.py:174-176); the only exchange is returning the token-id buffers to a root
rank: one count all-gather (8 bytes per rank) plus ONE grouped exchange of the u32 buffers at their exact
lengths (send / recv pairs: on RCCL one ncclGroup, the same calls the one-process entry
tk_group_encode_batch_device makes) -- 7 peers send to the root over 7 separate xGMI links, so nothing
is ring-serialised.  What the root's links can take bounds it: a rank's ids are about as many bytes as
its text (4 bytes per 4.2-byte token), so gathering 7 shards of 1 GiB moves 7 GB into one device per step.
"""
from __future__ import annotations

import numpy as np


def partition_by_bytes(doc_off: np.ndarray, world: int) -> list[tuple[int, int]]:
    """Split documents into `world` contiguous ranges of (nearly) equal bytes, preserving order.
    doc_off: uint64[n_docs+1].  Returns [(first_doc, last_doc_exclusive)] per rank."""
    doc_off = np.asarray(doc_off, dtype=np.uint64)
    n_docs = len(doc_off) - 1
    total = int(doc_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        d = int(np.searchsorted(doc_off, target, side="left"))
        cuts.append(min(max(d, cuts[-1]), n_docs))
    cuts.append(n_docs)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class PendingGather:
    """Handle of a token gather that may still be in flight (gather_tokens(..., async_op=True))."""

    def __init__(self, works, bufs, counts, keep):
        self._works, self._bufs, self.counts, self._keep = list(works or []), bufs, counts, keep

    def wait(self):
        """Blocks until the gather has completed.  Returns (per-rank tensors in rank order, counts) on the
        destination rank and (None, counts) elsewhere."""
        for w in self._works:
            w.wait()
        self._works = []
        self._keep = None
        if self._bufs is None:
            return None, self.counts
        return [b[:c] for b, c in zip(self._bufs, self.counts)], self.counts


def gather_tokens(tokens, n_tokens: int, rank: int, world: int, dist, torch, dst: int = 0, async_op: bool = False, padded: bool = False):
    """Gather per-rank token buffers (1-D integer tensors, first n_tokens entries valid) on rank `dst`.
    Returns (list of per-rank tensors in rank order, counts) on dst and (None, counts) elsewhere; with
    async_op=True returns a PendingGather instead, so that the transfer over xGMI overlaps with the encode of the
    next sub-batch (the counts are exchanged synchronously: 8 bytes per rank).  The buffers travel at their exact
    lengths (grouped send / recv); padded=True is the earlier form, one dist.gather of max(count) entries per rank.
    `tokens` is read until the gather has completed: the caller keeps it unchanged until wait() returns."""
    mine = torch.tensor([n_tokens], dtype=torch.int64, device=tokens.device)
    parts = [torch.zeros(1, dtype=torch.int64, device=tokens.device) for _ in range(world)]
    dist.all_gather(parts, mine)
    counts = [int(p.item()) for p in parts]
    if padded:
        cmax = max(max(counts), 1)
        if tokens.numel() >= cmax:
            send = tokens[:cmax].contiguous()
        else:
            send = torch.cat([tokens, tokens.new_zeros(cmax - tokens.numel())])
        bufs = [torch.empty(cmax, dtype=tokens.dtype, device=tokens.device) for _ in range(world)] if rank == dst else None
        work = dist.gather(send, bufs, dst=dst, async_op=async_op)
        pending = PendingGather([work] if async_op else [], bufs, counts, send)
        return pending if async_op else pending.wait()
    own = tokens[:n_tokens]
    ops, bufs = [], None
    if rank == dst:
        bufs = [own if r == dst else torch.empty(counts[r], dtype=tokens.dtype, device=tokens.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != dst and counts[r]]
    elif n_tokens:
        ops = [dist.P2POp(dist.isend, own.contiguous(), dst)]
    works = dist.batch_isend_irecv(ops) if ops else []
    pending = PendingGather(works, bufs, counts, own)
    if async_op:
        return pending
    return pending.wait()


def encode_ordinary_batch_sharded(encode_packed, blob: np.ndarray, doc_off: np.ndarray, rank: int, world: int, dist, torch,
                                  device="cpu"):
    """Doc-sharded encode: every rank encodes its contiguous byte-balanced range with
    `encode_packed(blob_slice, doc_off_slice) -> (tokens uint32[T], tok_off uint64[n+1])`, rank 0 gets
    (tokens, tok_off) of the whole batch in document order, other ranks get None."""
    first, last = partition_by_bytes(doc_off, world)[rank]
    a, b = int(doc_off[first]), int(doc_off[last])
    toks, toff = encode_packed(blob[a:b], (doc_off[first:last + 1] - doc_off[first]).astype(np.uint64))
    t = torch.from_numpy(np.ascontiguousarray(toks).view(np.int32)).to(device)
    parts, counts = gather_tokens(t, len(toks), rank, world, dist, torch)
    # per-document token counts travel the same way (tiny next to the ids)
    dcount = torch.from_numpy(np.diff(toff).astype(np.int64)).to(device)
    dparts, _ = gather_tokens(dcount, len(dcount), rank, world, dist, torch)
    if rank != 0:
        return None
    all_toks = np.concatenate([p.cpu().numpy().view(np.uint32) for p in parts]) if parts else np.zeros(0, np.uint32)
    all_counts = np.concatenate([p.cpu().numpy() for p in dparts]).astype(np.uint64)
    tok_off = np.zeros(len(all_counts) + 1, np.uint64)
    np.cumsum(all_counts, out=tok_off[1:])
    return all_toks, tok_off


def digest_algorithm() -> int:
    """Which function ids_digest uses in THIS interpreter: 1 = xxhash64 (SURVEY.md 8(d) names it for 1 GB+ comparisons), 2 = blake2b truncated
    to 8 bytes (the package is missing).  Travels with every verdict: ranks of one job that disagree are an error, not a "gather mismatch"."""
    try:
        import xxhash  # noqa: F401

        return 1
    except ImportError:  # pragma: no cover
        return 2


def ids_digest(ids) -> tuple[int, int]:
    """(count, 64-bit digest) of a flat id stream (any integer array of 4-byte items; the bytes are what is hashed); digest_algorithm() says
    with which function."""
    a = np.ascontiguousarray(np.asarray(ids)).view(np.uint8)
    if digest_algorithm() == 1:
        import xxhash

        return int(a.size // 4), int(xxhash.xxh64(memoryview(a)).intdigest())
    import hashlib  # pragma: no cover

    return int(a.size // 4), int.from_bytes(hashlib.blake2b(memoryview(a), digest_size=8).digest(), "little")  # pragma: no cover


def exchange_verdicts(n_tokens: int, digest: int, ok: bool, rank: int, world: int, dist, torch, device="cpu"):
    """Every rank tells every rank (count, digest of its own id stream, did its own shard equal the oracle's): one all-gather of
    five int64 per rank (the fifth: the digest function's id).  Returns [(count, digest, ok)] in rank order; RuntimeError when the ranks do
    not all digest with the same function (a job over mixed environments: every comparison of digests would be a false mismatch)."""
    algo = digest_algorithm()
    mine = torch.tensor([n_tokens, digest & 0xFFFFFFFF, digest >> 32, 1 if ok else 0, algo], dtype=torch.int64, device=device)
    parts = [torch.zeros(5, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, mine)
    out, algos = [], []
    for p in parts:
        c, lo, hi, k, a = (int(x) for x in p.cpu().tolist())
        out.append((c, (hi << 32) | lo, bool(k)))
        algos.append(a)
    if len(set(algos)) != 1:
        raise RuntimeError(f"the ranks digest their id streams with different functions ({algos}: 1 = xxhash64, 2 = blake2b): install the same packages on every rank")
    return out


def verify_gathered(parts, verdicts) -> dict:
    """On the destination rank: what was RECEIVED per peer (`parts`: the per-rank tensors / arrays a gather returned, rank order) against
    what the ranks say they sent (`verdicts`: exchange_verdicts()).  A gather that delivered one rank's ids twice, dropped a tail or
    mixed two buffers fails here; the ranks' own comparisons with the oracle make the received ids the oracle's by transitivity.
    Returns {"gather_verified": bool, "per_rank": [...]}."""
    per = []
    for r, (p, (cnt, dig, ok)) in enumerate(zip(parts, verdicts)):
        a = p.cpu().numpy() if hasattr(p, "cpu") else np.asarray(p)
        c, d = ids_digest(a)
        per.append({"rank": r, "tokens_sent": cnt, "tokens_received": c, "digest_equal": bool(c == cnt and d == dig), "shard_equal_to_oracle": ok})
    return {"gather_verified": bool(len(per) == len(verdicts) and all(x["digest_equal"] for x in per)), "per_rank": per}
