"""The merge schedule planned for the next kernels (tools/sim_merge_steps.py): K merges of byte_pair_merge (reference
src/lib.rs:140-196) per round of table probes -- the K merges chosen BEFORE any probe is answered, no two of them neighbours, carried out
in order as long as each is the lowest pair of the state it meets.  Checked here on the CPU: the tokens are the plain loop's on adversarial
vocabularies (ranks that do not grow with the length) and on the corpus' long pieces, and the probes of a carried-out merge were made with
the ids it meets (an assert inside the simulation)."""
import os
import random
import sys

import helpers as h

sys.path.insert(0, os.path.join(h.ROOT, "tools"))
import sim_merge_steps as sim  # noqa: E402


def test_k_merges_per_step_equal_the_plain_loop_on_adversarial_vocabularies():
    rng = random.Random(1)
    trials = 0
    for _ in range(1500):
        alpha = b"abc"[: rng.choice([2, 3])]
        V = {bytes([ch]): 1000 + ch for ch in alpha}
        toks = [bytes([c]) for c in alpha]
        for _ in range(rng.randint(3, 14)):  # tokens are concatenations of tokens; their ranks are shuffled: a longer token may rank below its parts' pair
            t = rng.choice(toks) + rng.choice(toks)
            if t not in V and len(t) <= 8 and t not in toks:
                toks.append(t)
        extra = [t for t in toks if len(t) > 1]
        rng.shuffle(extra)
        V.update({t: r for r, t in enumerate(extra)})
        sim.set_vocab(V)
        for _ in range(12):
            p = bytes(rng.choice(alpha) for _ in range(rng.randint(2, 40)))
            want = sim.reference(p)
            for K in (2, 3, 4, 8):
                steps, merges, parts = sim.steps_as_a_kernel_would(p, K)
                assert parts == want, (V, p, K)
                assert merges == len(p) - len(want) and steps <= max(merges, 1)
                assert sim.steps_lane_by_lane(p, K) == (steps, merges, parts), (V, p, K)  # (the kernel's bookkeeping, lane by lane: same rounds, same parts)
                trials += 1
    assert trials > 50000


def test_the_surveys_counterexample_for_merging_all_local_minima():
    """SURVEY.md 5 ("Giant pieces"): merging every local minimum of a round is NOT the reference's order; this schedule is."""
    V = {bytes([c]): 100 + c for c in b"abc"}
    V.update({t: r for r, t in enumerate([b"ac", b"acc", b"bb", b"acac", b"acb", b"bc", b"bbb"])})
    sim.set_vocab(V)
    p = b"caacbcbba"
    assert sim.reference(p) == [b"c", b"a", b"acb", b"c", b"bb", b"a"]
    for K in (2, 4, 8):
        assert sim.steps_as_a_kernel_would(p, K)[2] == sim.reference(p)


def test_k_merges_per_step_on_the_corpus_long_pieces():
    name = "o200k_shaped"
    V = h.golden_vocab(name)
    sim.set_vocab(V)
    C = h.c_oracle_for(name)
    blob, _ = h.gen_corpus(0x51D0C0, 1, 1 << 20)
    text = blob.tobytes()
    s, seen, one, four, merges = 0, set(), 0, 0, 0
    for e in C.split(text):
        p = text[s:e]
        s = e
        if len(p) < 25 or p in V or p in seen:
            continue
        seen.add(p)
        want = sim.reference(p)
        assert [t for q in want for t in [V[q]]] == C.encode_piece(p)  # (the simulation's plain loop is the oracle's)
        st4, mg4, parts = sim.steps_as_a_kernel_would(p, 4)
        assert parts == want
        if len(p) <= 256:
            assert sim.steps_lane_by_lane(p, 4) == (st4, mg4, parts)
        one += mg4
        four += st4
        merges += mg4
    assert len(seen) > 200
    assert one / four > 2.5, (one, four)  # (measured 2.9: profiles/r04_merge_steps_sim.txt)
