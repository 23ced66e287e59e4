#!/usr/bin/env python3
"""CPU simulation for SURVEY row "mergeable_ranks as an LDS-resident hash" inside the merge kernel (round 6 close-out): would an
LDS copy of the hottest entries of the PAIR table shorten tk_k_merge_all's steps?

Every merge of byte_pair_merge (src/lib.rs:140-196) asks the table for the two pairs the merged part makes with its neighbours, and a step of
the kernel waits for the slowest of its (two or four) probes -- so a step is shortened only if ALL its probes hit the LDS copy.  The
simulation replays the reference's merges over the distinct pieces of the web corpus that are not tokens (the merge kernel's work list),
records every pair probe (left id, right id), and reports, for hot sets of the K most frequently probed pairs: the share of probes they
answer, the share of merges BOTH of whose probes they answer, and of two-merge steps all four -- against what they cost: 8 bytes a pair, and
the kernel runs 16 wavefronts per CU on 40 KiB of LDS per workgroup of four (a hot set of 4 Ki pairs = 32 KiB per workgroup halves that).

usage: python tools/sim_pair_hotset.py [encoding] [MiB of corpus]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h

INF = 1 << 62


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "o200k_shaped"
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    V = h.load_vocab(name)
    C = h.c_oracle_for(name)
    blob, off = h.gen_corpus(0x5EED0003, 1, mib << 20)
    data = blob.tobytes()
    ends = C.split(data)
    pieces, a = collections.Counter(), 0
    for e in ends:
        p = data[a:e]
        a = e
        if 2 <= len(p) <= 1024 and p not in V:
            pieces[p] += 1
    print(f"{name}, {mib} MiB of the web corpus: {sum(pieces.values())} occurrences of pieces that are not tokens, {len(pieces)} distinct (the merge kernel's list)")
    probes = collections.Counter()
    per_merge = []  # the (one or two) probes of every merge, in merge order per piece: [(pair, ...), ...]
    n_merges = 0
    for p in pieces:
        parts = [p[i:i + 1] for i in range(len(p))]
        ranks = [V.get(parts[i] + parts[i + 1], INF) for i in range(len(parts) - 1)]
        seq = []
        while ranks:
            r = min(ranks)
            if r >= INF:
                break
            i = ranks.index(r)
            parts[i:i + 2] = [parts[i] + parts[i + 1]]
            del ranks[i]
            asked = []
            if i < len(parts) - 1:
                ranks[i] = V.get(parts[i] + parts[i + 1], INF)
                asked.append((parts[i], parts[i + 1]))
            if i > 0:
                ranks[i - 1] = V.get(parts[i - 1] + parts[i], INF)
                asked.append((parts[i - 1], parts[i]))
            for q in asked:
                probes[q] += 1
            seq.append(tuple(asked))
            n_merges += 1
        per_merge.append(seq)
    total = sum(probes.values())
    print(f"{n_merges} merges, {total} pair probes, {len(probes)} distinct pairs asked for ({sum(1 for q in probes if q[0] + q[1] in V)} of them are tokens)")
    order = [q for q, _ in probes.most_common()]
    print("hot set   LDS bytes   probes answered   merges with all probes answered   steps of two merges with all four answered")
    for K in (256, 1024, 4096, 16384, 65536):
        hot = set(order[:K])
        ans = sum(probes[q] for q in hot)
        m_all = m_tot = s_all = s_tot = 0
        for seq in per_merge:
            for asked in seq:
                m_tot += 1
                m_all += all(q in hot for q in asked)
            for j in range(0, len(seq) - 1, 2):
                s_tot += 1
                s_all += all(q in hot for q in seq[j] + seq[j + 1])
        print(f"{K:7d}  {K * 8 // 1024:6d} KiB   {100.0 * ans / total:6.1f} %          {100.0 * m_all / max(m_tot, 1):6.1f} %                          {100.0 * s_all / max(s_tot, 1):6.1f} %")


if __name__ == "__main__":
    main()
