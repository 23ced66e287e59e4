#!/usr/bin/env python3
"""CPU simulation: how many of the reference's merges (byte_pair_merge, src/lib.rs:140-196: always the leftmost pair of the lowest rank) fit
into ONE step of a merge kernel that prepares up to K merges side by side -- merge 1 is the lowest key; merge t is the lowest key among the
positions no earlier merge of the step has touched (its own part, the absorbed part, the part before it), and is carried out iff every new
pair the earlier merges of the step create ranks above it: then it IS the reference's next merge.  A step costs one round of table probes
(~1 us, profiles/r04_merge_steps.txt), whatever K is.  Output: merges per step for K = 1 .. 4 on the corpus' pieces that are not tokens,
by length class, and the tokens compared with the plain reference loop.

usage: python tools/sim_merge_steps.py [encoding] [MiB of corpus]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h

INF = 1 << 62
APART = os.environ.get("SIM_APART", "0") == "1"
KS = tuple(int(k) for k in os.environ.get("SIM_K", "1,2,3,4").split(","))


V = {}  # the vocabulary the functions below look pairs up in (bytes -> rank): set_vocab()


def set_vocab(v):
    global V
    V = v


def rank(a: bytes, b: bytes) -> int:
    return V.get(a + b, INF)


def reference(piece: bytes):
    parts = [piece[i:i + 1] for i in range(len(piece))]
    while len(parts) > 1:
        best, bi = INF, -1
        for i in range(len(parts) - 1):
            r = rank(parts[i], parts[i + 1])
            if r < best:
                best, bi = r, i
        if bi < 0:
            break
        parts[bi:bi + 2] = [parts[bi] + parts[bi + 1]]
    return parts


def steps_with(piece: bytes, K: int):
    """(steps, merges, parts) of the K-at-a-time schedule; the merges themselves are the reference's, one by one -- a step ends when the
    reference's next merge sits at a position the step has touched (a pair a merge of this step created) or after K merges."""
    parts = [piece[i:i + 1] for i in range(len(piece))]
    ident = list(range(len(piece)))  # a stable name per part (its first byte's position), to track "touched"
    steps = merges = 0
    while len(parts) > 1:
        touched = set()
        done = 0
        while done < K and len(parts) > 1:
            best, bi = INF, -1
            for i in range(len(parts) - 1):
                r = rank(parts[i], parts[i + 1])
                if r < best:
                    best, bi = r, i
            if bi < 0:
                break
            if done and (ident[bi] in touched or (APART and ident[bi + 1] in touched)):
                break  # the reference's next merge starts at a part whose pair this step has made (a merged part, or the part before one): it needs the probes' answers first
            # the merge touches: its own part (new pair with the part behind), the absorbed part, the part before (new pair with the merged part)
            touched.add(ident[bi]); touched.add(ident[bi + 1])
            if bi > 0:
                touched.add(ident[bi - 1])
            # SIM_APART=1: ... and the part behind the absorbed one, and a merge may not absorb a touched part either: no two merges of a step are
            # neighbours, so the operands of all 2K probes are known when the K merges are chosen (no probe depends on which of them are carried out)
            if APART and bi + 2 < len(parts):
                touched.add(ident[bi + 2])
            parts[bi:bi + 2] = [parts[bi] + parts[bi + 1]]
            del ident[bi + 1]
            done += 1
        if not done:
            break
        steps += 1
        merges += done
    return steps, merges, parts


def steps_as_a_kernel_would(piece: bytes, K: int):
    """The same schedule the way a kernel has to do it (SIM_APART's rule): the K merges of a step are CHOSEN before any probe is answered --
    t-th choice: the lowest pair whose two parts no earlier choice has touched (touched: the part before, the merged part, the absorbed part, the
    part behind) --, all 2K new pairs are probed at once with the parts' ids as they were BEFORE the step, and then the choices are carried out
    in order as long as each is the lowest pair of the state it meets (otherwise the step ends: the rest is chosen again next step).
    Asserts that a carried-out merge's probes were made with the ids it meets.  Returns (steps, merges, parts)."""
    n = len(piece)
    ids = [piece[i:i + 1] for i in range(n)]
    nx = list(range(1, n + 1))  # n = none
    pv = [-1] + list(range(n - 1))
    key = [(rank(ids[i], ids[i + 1]), i) if i + 1 < n else (INF, i) for i in range(n)]
    alive = [True] * n
    steps = merges = 0
    while True:
        touched, chosen = set(), []
        for _ in range(K):
            best = (INF, -1)
            for i in range(n):
                if alive[i] and nx[i] < n and i not in touched and nx[i] not in touched and key[i] < best:
                    best = key[i]
            if best[0] >= INF:
                break
            i = best[1]
            j = nx[i]
            chosen.append(i)
            touched.update((i, j))
            if pv[i] >= 0:
                touched.add(pv[i])
            if nx[j] < n:
                touched.add(nx[j])
        if not chosen:
            break
        # the probes, all of them with the ids of before the step
        probed = {}
        for i in chosen:
            j = nx[i]
            m = ids[i] + ids[j]
            nn, pp = nx[j], pv[i]
            probed[i] = (m, ids[nn] if nn < n else None, ids[pp] if pp >= 0 else None,
                         rank(m, ids[nn]) if nn < n else INF, rank(ids[pp], m) if pp >= 0 else INF)
        done = 0
        for i in chosen:
            if min(key[k] for k in range(n) if alive[k]) != key[i]:
                break  # a pair made by this step (or one beside a merge of it) ranks lower: the reference merges that one first
            m, idn, idp, r_i, r_p = probed[i]
            j = nx[i]
            nn, pp = nx[j], pv[i]
            assert (ids[nn] if nn < n else None) == idn and (ids[pp] if pp >= 0 else None) == idp and ids[i] + ids[j] == m
            ids[i] = m
            alive[j] = False
            nx[i] = nn
            if nn < n:
                pv[nn] = i
            key[i] = (r_i, i)
            if pp >= 0:
                key[pp] = (r_p, pp)
            done += 1
        assert done >= 1
        steps += 1
        merges += done
    return steps, merges, [ids[i] for i in range(n) if alive[i]]



def steps_lane_by_lane(piece: bytes, K: int):
    """tk_k_small's sixteen-lane merge with K merges per round (round 4: written for tk_fused.h behind -DTK_SMALL_ONE_PHASE=2; measured in round 5 -- slower than two merges per round, profiles/r05_small_variants.txt -- and removed there) transliterated lane by lane: the arrays at
    the piece's positions (id, rk, nx, pv, the round stamps st), what each of the sixteen lanes scans (positions g, g + 16, ...), which lane
    probes and writes what, in the kernel's order of statements.  A check of the kernel's bookkeeping, not only of the schedule.
    Returns (rounds, merges, parts)."""
    n, G, MAXR, NONE = len(piece), 16, INF, 0xFFFF
    id_ = [piece[i:i + 1] for i in range(n)]
    rk = [rank(id_[k], id_[k + 1]) if k + 1 < n else MAXR for k in range(n)]
    nx = [k + 1 for k in range(n)]
    pv = [k - 1 if k else NONE for k in range(n)]
    st = [0] * n
    rounds = merges = 0
    rnd = 0
    while True:
        rnd += 1
        ci, cm, nc, more = [None] * K, [MAXR] * K, 0, True
        for t in range(K):
            br, bk = [MAXR] * G, [None] * G
            if more:
                for g in range(G):
                    for k in range(g, n, G):
                        r = rk[k]
                        if r < br[g] and st[k] != rnd and st[nx[k]] != rnd:
                            br[g], bk[g] = r, k
            m = min(br)                                                    # tkm_group_min(br, 4)
            have = m != MAXR
            i = min([bk[g] for g in range(G) if br[g] == m and bk[g] is not None], default=None)
            if have:
                ci[t], cm[t], nc = i, m, t + 1
                j = nx[i]; nn = nx[j]; pp = pv[i]                          # lane 0 of the piece
                st[i] = rnd; st[j] = rnd
                if pp != NONE: st[pp] = rnd
                if nn < n: st[nn] = rnd
            more = have
        if nc == 0:
            break
        # the probes: lane 2t the right pair, lane 2t + 1 the left pair of choice t -- with the arrays as they are BEFORE any commit
        newr, loc = [MAXR] * G, [None] * G
        for g in range(G):
            tt = g >> 1
            if tt >= K or ci[tt] is None:
                continue
            pi, pm = ci[tt], cm[tt]
            pj = nx[pi]; pnn = nx[pj]; ppp = pv[pi]
            merged = id_[pi] + id_[pj]
            assert V.get(merged, INF) == pm
            right = not (g & 1)
            loc[g] = (pi, pj, pnn, ppp, merged)
            if right and pnn < n: newr[g] = rank(merged, id_[pnn])
            if not right and ppp != NONE: newr[g] = rank(id_[ppp], merged)
        done = 0
        for t in range(K):
            go = t < nc
            if t > 0 and go:
                best = min(((rk[k], k) for k in range(n)), default=(MAXR, None))   # the sixteen lanes' scans + two reductions
                go = best[0] == cm[t] and best[1] == ci[t]
                if not go: nc = t
            if go:
                pi, pj, pnn, ppp, merged = loc[2 * t]
                assert nx[pi] == pj and nx[pj] == pnn and pv[pi] == ppp and id_[pi] + id_[pj] == merged  # (the probes' operands are what the merge meets)
                id_[pi] = merged; nx[pi] = pnn                               # lane 2t
                if pnn < n: pv[pnn] = pi
                rk[pj] = MAXR; id_[pj] = None; rk[pi] = newr[2 * t]
                if ppp != NONE: rk[ppp] = newr[2 * t + 1]                   # lane 2t + 1
                done += 1
        rounds += 1
        merges += done
    return rounds, merges, [x for x in id_ if x is not None]


def main():
    global APART, steps_with
    name = sys.argv[1] if len(sys.argv) > 1 else "o200k_shaped"
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    set_vocab(h.golden_vocab(name))
    C = h.c_oracle_for(name)
    blob, _ = h.gen_corpus(0x51D0C0, 1, mib << 20)
    text = blob.tobytes()
    KERNEL_FORM = os.environ.get("SIM_KERNEL_FORM", "0") == "1"
    if KERNEL_FORM:
        APART = True
        steps_with = steps_as_a_kernel_would
    classes = [(25, 32), (33, 48), (49, 64), (65, 128), (129, 256), (257, 1 << 30)]
    if os.environ.get("SIM_SHORT"):  # the pieces of the one-lane merges instead
        classes = [(2, 4), (5, 8), (9, 16), (17, 24)]
    acc = {K: {c: [0, 0] for c in classes} for K in KS}
    pieces = {c: 0 for c in classes}
    s, bad, seen = 0, 0, set()
    for e in C.split(text):
        p = text[s:e]
        s = e
        if len(p) < classes[0][0] or len(p) > classes[-1][1] or p in V or p in seen:
            continue
        seen.add(p)
        c = next(c for c in classes if c[0] <= len(p) <= c[1])
        pieces[c] += 1
        ref = reference(p)
        for K in acc:
            st, mg, parts = steps_with(p, K)
            bad += parts != ref
            acc[K][c][0] += st
            acc[K][c][1] += mg
    print(f"{name}, {mib} MiB of web text{' (merges of a step kept apart: SIM_APART=1)' if APART else ''}{' [chosen before the probes, as a kernel has to: SIM_KERNEL_FORM=1]' if KERNEL_FORM else ''}: distinct pieces of 25 bytes and more that are not tokens; merges per step (steps) for K merges prepared side by side; mismatches against the plain loop: {bad}")
    print("bytes        pieces   merges   " + "   ".join(f"K={K}" + " " * 13 for K in acc))
    for c in classes:
        if not pieces[c]:
            continue
        row = f"{c[0]:4d}-{min(c[1], 9999):<5d} {pieces[c]:7d} {acc[1][c][1]:8d}   "
        row += "   ".join(f"{acc[K][c][1] / max(acc[K][c][0], 1):4.2f} ({acc[K][c][0]:7d})" for K in acc)
        print(row)
    tot = {K: [sum(acc[K][c][0] for c in classes), sum(acc[K][c][1] for c in classes)] for K in acc}
    print(f"all        {sum(pieces.values()):7d} {tot[1][1]:8d}   " + "   ".join(f"{tot[K][1] / max(tot[K][0], 1):4.2f} ({tot[K][0]:7d})" for K in acc))


if __name__ == "__main__":
    main()
