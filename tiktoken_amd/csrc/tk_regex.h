// The generic pat_str engine: any split pattern outside the three hand-written scanner families (tk_pattern.cpp) is compiled once per
// Encoding (reference: Regex::new(pattern), src/lib.rs:623) into a small backtracking program that the GPU runs itself
// (tk_regex_kernels.h).  This header holds what host and device share: the program layout, the property lookup and the matcher.
//
// Semantics are those of fancy-regex / Python `regex` for the supported syntax (tk_regex.cpp): leftmost, alternatives in order,
// greedy / lazy / possessive quantifiers, atomic groups, look-ahead, case-insensitive literals, \b and look-behind of fixed length; no
// back-references.  The end of the piece that starts at p depends only on the TEXT around p -- from p on, plus a few chars before a
// position for \b and (?<=X) -- and on where its haystack begins and ends, never on where earlier pieces began: that is what lets the
// kernels evaluate piece starts speculatively in parallel and prove them afterwards (tk_regex_kernels.h).
//
// A program is an array of 16-byte instructions over "sets" (sets of code points).  Repeats of a single set -- \s+, \p{L}*, [^\r\n]+? --
// are ONE instruction with ONE backtrack frame however long the run is (a frame holds a range of positions), and a repeated GROUP that is
// possessive, or stands where nothing can fail behind it (the tail of an alternative, of a look-ahead, of an atomic group), forgets each
// repetition's alternatives as it goes: the backtrack stack is bounded by the nesting of the pattern, not by the text, for everything but
// a backtracking repeated group in the middle of an alternative ((?:ab)*c), which is good for TK_RX_STACK / 2 repetitions.
#pragma once
#include <stdint.h>

#include "tk_common.h"

enum {
    TK_RX_SET = 0,     // a: set -- one char of the set
    TK_RX_REP,         // a: set, b: min, c: max (0xFFFFFFFF = unbounded); mode = (op >> 8) & 3
    TK_RX_SPLIT,       // a: first choice, b: second choice; (op >> 8) & 127 and (op >> 16) & 127: 1 + index of the first-byte bitmap of
                       // each choice (0: none) -- a choice whose bitmap lacks the byte at the position is not taken, nor kept as a way back
    TK_RX_JMP,         // a: target
    TK_RX_MATCH,
    TK_RX_END,         // end of the haystack ($, \z)
    TK_RX_START,       // start of the haystack (^, \A)
    TK_RX_ATOM_BEGIN,  // (?>...), possessive quantifiers of groups
    TK_RX_ATOM_END,
    TK_RX_LOOK_BEGIN,  // a: 1 = negative, b: instruction behind the look-ahead
    TK_RX_LOOK_END,
    TK_RX_FAIL,
    TK_RX_POP,  // forget the newest alternative (possessive loops of groups: the way out of the previous repetition)
    TK_RX_WORDB,  // a: 0 = \b, 1 = \B -- word boundary between the char before the position and the char at it
    TK_RX_PREV,   // a: set, c: distance d >= 1 -- look-behind: the d-th char before the position exists (in this haystack) and is in the set
};
enum { TK_RX_GREEDY = 0, TK_RX_LAZY = 1, TK_RX_POSSESSIVE = 2 };
#define TK_RX_INF 0xFFFFFFFFu

struct TkRxIns {  // 16 bytes
    uint32_t op, a, b, c;
};
// A set of code points: ASCII members as a bitmap (everything below already applied); beyond ASCII the union of General_Category members
// (gcmask, bit = index in tools/gen_regex_props.py's order), \s / \w members (flags bits 5 / 6, the bits of the property byte), explicit
// ranges and -- flags bit 1 -- the COMPLEMENT of one more such term (cgcmask, flags bits 13 / 14: the \S of [^\S\n]); flags bits 16..22:
// 1 + the set the members must ALSO be in ([A&&B], [A--B]; 0 = none; that set chains on the same way); flags bit 0 negates the whole.
struct TkRxSet {  // 32 bytes
    uint32_t ascii[4];
    uint32_t gcmask, cgcmask;
    uint32_t flags;
    uint32_t rr;  // ranges[roff .. roff + rcnt): pairs (lo, hi), inclusive; roff << 16 | rcnt
};
struct TkRxProg {
    const TkRxIns* ins;
    const TkRxSet* sets;
    const uint32_t* ranges;
    const uint8_t* stage1;  // [0x1100] property table (tk_regex_props.inc)
    const uint8_t* stage2;
    uint32_t n_ins, n_sets, n_ranges;
    const uint32_t* first;  // first-byte bitmaps, 8 words each: the bytes with which a match of the rest of the program from some
    uint32_t n_first;       // instruction can begin (tk_regex.cpp: a fixpoint over the program; conservative, so skipping is exact)
    // The same pattern as a DFA (tk_regex_dfa.inc; null: the pattern has none -- look-behind, \b, general atomic groups -- and the program
    // above runs).  trans[state * ncls + cls]: bit 15 = "a match ends HERE, in front of this char", bits 0..14 = the next state (0: dead).
    // State 1 starts a match at the first char of its haystack (^, \A), state 1 + g one behind a char of group g (one group, unless the
    // pattern looks behind: dfa_flags bit 0, groups in dfa_ascii[128 + class]).  cls 0 is the end of the haystack, 1 .. ncls - 1 the
    // classes of code points no set of the pattern tells apart: ASCII through dfa_ascii[128], the rest through the two-stage table.
    const uint16_t* dfa_trans = nullptr;
    const uint8_t* dfa_ascii = nullptr;
    const uint16_t* dfa_s1 = nullptr;  // [0x1100]: bit 15 set -> the class of all 256 code points (CJK, Hangul, unassigned planes: no second look-up); else block of dfa_s2
    const uint8_t* dfa_s2 = nullptr;   // blocks of 256 classes
    uint32_t dfa_ncls = 0, dfa_flags = 0;
};

#define TK_RX_FAILED 0xFFFFFFFFu    // no match at this position
#define TK_RX_OVERFLOW 0xFFFFFFFEu  // backtrack stack exhausted (a repeated group on a long text)
#define TK_RX_LIMIT 0xFFFFFFFDu     // backtrack budget exhausted (nested quantifiers that explode: the reference's fancy-regex gives up too)
#define TK_RX_IS_ERROR(q) ((q) >= TK_RX_LIMIT)
// Work one match may do, in steps (an instruction, a char of a repeat, a backtrack): fancy-regex's default limit of 1 000 000 backtracks
// (Error::BacktrackLimitExceeded, which the reference turns into a panic) plus what linear walks over the text itself need (\s*[\r\n]+ on
// a megabyte of blanks runs forward once and steps back a char at a time).
#define TK_RX_BUDGET_BASE 1000000u
#define TK_RX_BUDGET_PER_BYTE 8u
#define TK_RX_STACK 64
#ifndef TK_RX_ON_DONE
#define TK_RX_ON_DONE(steps)  // (the CPU tests add up the matcher's work here)
#endif

TK_HD uint32_t tk_rx_prop(const TkRxProg& P, uint32_t cp) {
    if (cp > 0x10FFFFu) cp = 0xFFFDu;
    return P.stage2[(uint32_t)P.stage1[cp >> 8] * 256u + (cp & 255u)];
}

// \w: Alphabetic | M | Nd | Pc | Join_Control (bit 6 of the property byte)
TK_HD bool tk_rx_is_word(const TkRxProg& P, uint32_t cp) { return (tk_rx_prop(P, cp) & 0x40u) != 0u; }

// membership of a non-ASCII code point before negation and intersection (pr: its property byte)
TK_HD bool tk_rx_raw_member(const TkRxProg& P, const TkRxSet& S, uint32_t cp, uint32_t pr) {
    bool in = ((S.gcmask >> (pr & 31u)) & 1u) || (pr & S.flags & 0x60u);
    if (!in && (S.flags & 2u)) in = !(((S.cgcmask >> (pr & 31u)) & 1u) || (pr & (S.flags >> 8) & 0x60u));
    if (!in && (S.rr & 0xFFFFu)) {  // the ranges of a set are sorted and disjoint (tk_regex.cpp): first range that ends at or behind cp
        const uint32_t roff = S.rr >> 16;
        uint32_t lo = 0, hi = S.rr & 0xFFFFu;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cp > P.ranges[2 * (roff + mid) + 1]) lo = mid + 1;
            else hi = mid;
        }
        in = lo < (S.rr & 0xFFFFu) && cp >= P.ranges[2 * (roff + lo)];
    }
    return in;
}

TK_HD bool tk_rx_in_set(const TkRxProg& P, uint32_t s, uint32_t cp) {
    const TkRxSet& S = P.sets[s];
    if (cp < 128u) return (S.ascii[cp >> 5] >> (cp & 31u)) & 1u;
    const uint32_t pr = tk_rx_prop(P, cp);
    bool in = tk_rx_raw_member(P, S, cp, pr);
    for (uint32_t nx = (S.flags >> 16) & 127u; in && nx;) {  // [A&&B], [A--B]: the operands, each a set of its own (negated for --)
        const TkRxSet& B = P.sets[nx - 1u];
        in = tk_rx_raw_member(P, B, cp, pr) != (bool)(B.flags & 1u);
        nx = (B.flags >> 16) & 127u;
    }
    return in != (bool)(S.flags & 1u);
}

// The char at pos (pos < t.n): code point and length.  Malformed UTF-8 reads as U+FFFD, one byte long (the boundary's contract is valid
// UTF-8; this only keeps the walk inside the text).
template <class A>
TK_HD uint32_t tk_rx_decode(A& t, uint32_t pos, uint32_t* len) {
    const uint32_t b0 = t.byte(pos);
    *len = 1;
    if (b0 < 0x80u) return b0;
    const uint32_t need = b0 >= 0xF0u ? 4u : (b0 >= 0xE0u ? 3u : (b0 >= 0xC0u ? 2u : 0u));
    if (!need || (uint64_t)pos + need > t.n) return 0xFFFDu;
    uint32_t cp = b0 & (0x7Fu >> need);
    for (uint32_t i = 1; i < need; ++i) {
        const uint32_t b = t.byte(pos + i);
        if ((b & 0xC0u) != 0x80u) return 0xFFFDu;
        cp = (cp << 6) | (b & 0x3Fu);
    }
    *len = need;
    return cp;
}

// The DFA form of the matcher: one table look-up per char, no stack, no budget -- every lane of a wavefront runs the same loop whatever
// alternative of the pattern its text is in.  Leftmost-first semantics are in the table (tk_regex_dfa.inc: a state is an ORDERED list of
// NFA states, a match cuts off everything of lower priority, assertions about the next char are resolved by the class of that char), so
// the end of the match is the last position at which a transition said "match": exactly what tk_rx_match returns for the same pattern.
// class of a code point beyond ASCII
TK_HD uint32_t tk_rx_dfa_cls(const TkRxProg& P, uint32_t cp) {
    if (cp > 0x10FFFFu) cp = 0xFFFDu;
    const uint32_t e = P.dfa_s1[cp >> 8];
    return (e & 0x8000u) ? (e & 0xFFu) : P.dfa_s2[e * 256u + (cp & 255u)];
}
// class of an ASCII byte.  (Device: read as a word of the table in LDS -- a byte load here would be merged with the byte load of the
// non-ASCII path, which reads global memory, into one load through a generic pointer.)
TK_HD uint32_t tk_rx_ascii_cls(const TkRxProg& P, uint32_t b0) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (((const uint32_t*)P.dfa_ascii)[b0 >> 2] >> (8u * (b0 & 3u))) & 0xFFu;
#else
    return P.dfa_ascii[b0];
#endif
}

// the state in which a match that starts at `start` begins: 1 at the first char of a haystack, else 1 + the group of the char in front
// (PREV: the pattern looks behind -- dfa_flags bit 0 -- and the char is read, as TK_RX_PREV reads it: a malformed char before the position
// counts as U+FFFD.  A compile-time choice: the look-up costs the speculative kernel of every OTHER pattern 12 % when it is compiled in,
// profiles/r03_generic_engine.txt)
template <bool PREV, class A>
TK_HD uint32_t tk_rx_dfa_start(const TkRxProg& P, A& t, uint32_t start) {
    if (start == 0u || t.hard(start)) return 1u;
    if constexpr (!PREV) return 2u;
    uint32_t r = start - 1u, len;
    for (int k = 0; k < 3 && r > 0u && (t.byte(r) & 0xC0u) == 0x80u && !t.hard(r); ++k) --r;
    uint32_t cp = tk_rx_decode(t, r, &len);
    if (r + len != start) cp = 0xFFFDu;
    const uint32_t cls = cp < 0x80u ? tk_rx_ascii_cls(P, cp) : tk_rx_dfa_cls(P, cp);
    return 1u + tk_rx_ascii_cls(P, 128u + cls);
}

template <bool PREV, class A>
TK_HD uint32_t tk_rx_match_dfa(const TkRxProg& P, A& t, uint32_t start) {
    const uint32_t ncls = P.dfa_ncls;
    uint32_t state = tk_rx_dfa_start<PREV>(P, t, start);
    uint32_t pos = start, last = TK_RX_FAILED;
    for (;;) {
        uint32_t cls = 0u, len = 0u;  // (the end of the haystack)
        if (pos < t.n && !(pos > start && t.hard(pos))) {
            const uint32_t b0 = t.byte(pos);
            if (b0 < 0x80u) {
                cls = tk_rx_ascii_cls(P, b0);
                len = 1u;
            } else {
                cls = tk_rx_dfa_cls(P, tk_rx_decode(t, pos, &len));
            }
        }
        const uint32_t e = P.dfa_trans[state * ncls + cls];
        if (e & 0x8000u) last = pos;
        state = e & 0x7FFFu;
        if (state == 0u) break;  // (always behind the end of the haystack: nothing consumes it)
        pos += len;
    }
    TK_RX_ON_DONE(pos - start + 1u);
    return last;
}

// ---- One piece matched by a GROUP of lanes (the resolving pass: a wavefront per document stands on a true piece start, all of its lanes
// with the same arguments).  A long piece is, but for a few chars, a long run in ONE state that loops to itself -- \p{L}+ inside a word of
// a megabyte, \s+ inside blank lines -- so when the table walk below has stayed in one state for TK_RX_COOP_STREAK (8) chars the group scans
// ahead together: lane j takes the 16-byte block j of the next KiB (one aligned 16-byte load per lane: 1 KiB per load instruction, coalesced),
// finds the first position in it at which the run cannot go on (a char whose transition leaves the state, the end of the haystack, bytes
// that are not well-formed UTF-8: everything a lane cannot judge on its own is left to the walk) and the last position at which a
// transition flags a match; the minimum / maximum over the lanes say where the walk goes on.  Exact: the scan accepts only what the walk
// would do char by char.
#define TK_RX_COOP_LANES 64u
#define TK_RX_COOP_STREAK 8u
#define TK_RX_NONE 0xFFFFFFFFu

// lane's block: the 16 bytes at blk (16-byte aligned, < t.n).  *bad: first position >= pos in it where the run in state S ends, or NONE;
// *mat: last position in front of that at which the transition flags a match, or NONE; *endp: where the last char it accepted ends (a
// char that begins in the last block of a scan may end in the next KiB: the scan goes on behind it).
template <class A>
TK_HD void tk_rx_run_lane(const TkRxProg& P, A& t, uint32_t S, uint32_t start, uint32_t pos, uint32_t blk, uint32_t* bad, uint32_t* mat, uint32_t* endp) {
    *bad = *mat = TK_RX_NONE;
    *endp = blk + 16u;
    if (blk >= t.n) {
        *bad = blk > pos ? blk : pos;
        return;
    }
    uint32_t w[6];  // bytes blk - 4 .. blk + 19: the chars that reach into the block or out of it
    w[0] = blk ? t.word(blk - 4u) : 0u;
    t.block16(blk, w + 1);
    w[5] = t.word(blk + 16u);  // (the text is readable 64 bytes past n)
    const uint32_t hb = t.hard16(blk);
    const uint32_t ncls = P.dfa_ncls;
    bool done = false;
#define TK_RX_B(i) ((w[((i) + 4) >> 2] >> (8u * (((i) + 4) & 3u))) & 0xFFu)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t q = blk + (uint32_t)i;
        if (done || q < pos) continue;
        if (q >= t.n || (q > start && ((hb >> i) & 1u))) {
            *bad = q;
            done = true;
            continue;
        }
        const uint32_t b = TK_RX_B(i);
        uint32_t cls, len = 1u;
        if (b < 0x80u) {
            cls = tk_rx_ascii_cls(P, b);
        } else if (b < 0xC0u) {
            // a continuation byte: part of a char that begins at most three bytes earlier (at or behind pos) -- that char's own position
            // answers for it; anything else is a stray byte, which the walk reads as a char of its own
            bool covered = false;
            const uint32_t b1 = TK_RX_B(i - 1), b2 = TK_RX_B(i - 2), b3 = TK_RX_B(i - 3);
            if (q >= pos + 1u && b1 >= 0xC0u) covered = true;                                              // (every lead takes at least one)
            else if (q >= pos + 2u && b1 >= 0x80u && b1 < 0xC0u && b2 >= 0xE0u) covered = true;
            else if (q >= pos + 3u && b1 >= 0x80u && b1 < 0xC0u && b2 >= 0x80u && b2 < 0xC0u && b3 >= 0xF0u) covered = true;
            if (!covered) {
                *bad = q;
                done = true;
            }
            continue;
        } else {
            const uint32_t need = b >= 0xF0u ? 4u : (b >= 0xE0u ? 3u : 2u);
            const uint32_t c1 = TK_RX_B(i + 1), c2 = TK_RX_B(i + 2), c3 = TK_RX_B(i + 3);
            bool ok = (uint64_t)q + need <= t.n && (c1 & 0xC0u) == 0x80u;
            uint32_t cp = ((b & (0x7Fu >> need)) << 6) | (c1 & 0x3Fu);
            if (need >= 3u) {
                ok = ok && (c2 & 0xC0u) == 0x80u;
                cp = (cp << 6) | (c2 & 0x3Fu);
            }
            if (need == 4u) {
                ok = ok && (c3 & 0xC0u) == 0x80u;
                cp = (cp << 6) | (c3 & 0x3Fu);
            }
            // (a hard start inside the char cannot be: it would cut a char in two; left to the walk all the same)
            if (!ok || (hb >> (i + 1)) & ((1u << (need - 1u)) - 1u) & 0xFFFFu) {
                *bad = q;
                done = true;
                continue;
            }
            cls = tk_rx_dfa_cls(P, cp);
            len = need;
        }
        const uint32_t e = P.dfa_trans[S * ncls + cls];
        if ((e & 0x7FFFu) != S) {
            *bad = q;
            done = true;
            continue;
        }
        if (e & 0x8000u) *mat = q;
        *endp = q + len;
    }
#undef TK_RX_B
}

// coop(S, start, pos, base, &pbad, &m1, &pnext): every lane j of the group runs tk_rx_run_lane on the block base + 16 j; pbad = the smallest
// `bad` (NONE: the whole KiB is in the run -- pnext = the last lane's `endp` is where it goes on), m1 = 1 + the largest `mat` in front of pbad (0: none).
template <bool PREV, class A, class Coop>
TK_HD uint32_t tk_rx_match_dfa_coop(const TkRxProg& P, A& t, uint32_t start, Coop&& coop) {
    const uint32_t ncls = P.dfa_ncls;
    uint32_t state = tk_rx_dfa_start<PREV>(P, t, start);
    uint32_t pos = start, last = TK_RX_FAILED, streak = 0u;
    for (;;) {
        uint32_t cls = 0u, len = 0u;  // (the end of the haystack)
        if (pos < t.n && !(pos > start && t.hard(pos))) {
            const uint32_t b0 = t.byte(pos);
            if (b0 < 0x80u) {
                cls = tk_rx_ascii_cls(P, b0);
                len = 1u;
            } else {
                cls = tk_rx_dfa_cls(P, tk_rx_decode(t, pos, &len));
            }
        }
        const uint32_t e = P.dfa_trans[state * ncls + cls];
        if (e & 0x8000u) last = pos;
        const uint32_t nx = e & 0x7FFFu;
        if (nx == 0u) break;
        streak = nx == state ? streak + 1u : 0u;
        state = nx;
        pos += len;
        if (streak >= TK_RX_COOP_STREAK && t.limit == 0xFFFFFFFFu) {  // a run: the group scans ahead, a KiB per step
            for (;;) {
                const uint32_t base = pos & ~15u;
                uint32_t pbad, m1, pnext;
                coop(state, start, pos, base, &pbad, &m1, &pnext);
                if (m1) last = m1 - 1u;
                if (pbad != TK_RX_NONE) {
                    pos = pbad;
                    break;
                }
                pos = pnext;
            }
            streak = 0u;
        }
    }
    TK_RX_ON_DONE(pos - start + 1u);
    return last;
}

// End of the match of P that starts at `start`, TK_RX_FAILED or TK_RX_OVERFLOW.  `t` gives the text: byte(pos), n, hard(pos) -- a
// position where a new haystack begins (document start, special-token edge): the match sees end-of-text there, exactly like the slice of
// src/lib.rs:405.  Positions are 32-bit (a chunk is < 3 GiB).
template <class A>
TK_HD uint32_t tk_rx_match(const TkRxProg& P, A& t, uint32_t start) {
    enum { F_ALT = 0, F_RANGE = 1, F_LAZY = 2, F_ATOM = 3, F_LOOK = 4 };
    uint32_t fk[TK_RX_STACK], fp[TK_RX_STACK], fa_[TK_RX_STACK];  // frames: kind << 24 | pc, position, aux
    int sp = 0;
    uint32_t pc = 0, pos = start;
    uint32_t steps = 0, far = start;  // work done; the farthest position looked at
    auto spent = [&]() -> bool {
        const uint32_t span = far - start < (1u << 28) ? far - start : (1u << 28);
        return steps > TK_RX_BUDGET_BASE + TK_RX_BUDGET_PER_BYTE * span;
    };
    auto at_end = [&](uint32_t p) -> bool { return p >= t.n || (p > start && t.hard(p)); };
    for (;;) {
        const TkRxIns I = P.ins[pc];
        bool fail = false;
        if (pos > far) far = pos;
        ++steps;
        if (spent()) return TK_RX_LIMIT;
        switch (I.op & 0xFFu) {
            case TK_RX_SET: {
                uint32_t len;
                if (!at_end(pos) && tk_rx_in_set(P, I.a, tk_rx_decode(t, pos, &len))) {
                    pos += len;
                    ++pc;
                } else {
                    fail = true;
                }
            } break;
            case TK_RX_REP: {
                const uint32_t mode = (I.op >> 8) & 3u, mn = I.b, mx = I.c;
                const uint32_t want = mode == TK_RX_LAZY ? mn : mx;
                uint32_t c = 0, pmin = pos;
                while (c < want && !at_end(pos)) {
                    uint32_t len;
                    if (!tk_rx_in_set(P, I.a, tk_rx_decode(t, pos, &len))) break;
                    pos += len;
                    if (++c == mn) pmin = pos;
                }
                steps += c;
                if (pos > far) far = pos;
                if (c < mn) {
                    fail = true;
                    break;
                }
                if ((mode == TK_RX_GREEDY && pos > pmin) || (mode == TK_RX_LAZY && c < mx)) {
                    if (sp == TK_RX_STACK) return TK_RX_OVERFLOW;
                    fk[sp] = mode == TK_RX_GREEDY ? ((uint32_t)F_RANGE << 24 | (pc + 1)) : ((uint32_t)F_LAZY << 24 | pc);
                    fp[sp] = mode == TK_RX_GREEDY ? pmin : pos;
                    fa_[sp] = mode == TK_RX_GREEDY ? pos : c;
                    ++sp;
                }
                ++pc;
            } break;
            case TK_RX_SPLIT: {
                uint32_t go = 3u;  // bit 0: the first choice can begin with the byte here, bit 1: the second can
                const uint32_t fa = (I.op >> 8) & 127u, fb = (I.op >> 16) & 127u;
                if ((fa | fb) && !at_end(pos)) {
                    const uint32_t b = t.byte(pos);
                    if (fa && !((P.first[(fa - 1u) * 8u + (b >> 5)] >> (b & 31u)) & 1u)) go &= ~1u;
                    if (fb && !((P.first[(fb - 1u) * 8u + (b >> 5)] >> (b & 31u)) & 1u)) go &= ~2u;
                }
                if (go == 3u) {
                    if (sp == TK_RX_STACK) return TK_RX_OVERFLOW;
                    fk[sp] = (uint32_t)F_ALT << 24 | I.b;
                    fp[sp] = pos;
                    fa_[sp] = 0;
                    ++sp;
                    pc = I.a;
                } else if (go) {
                    pc = go == 1u ? I.a : I.b;
                } else {
                    fail = true;
                }
            } break;
            case TK_RX_JMP: pc = I.a; break;
            case TK_RX_MATCH: TK_RX_ON_DONE(steps); return pos;
            case TK_RX_END:
                if (at_end(pos)) ++pc;
                else fail = true;
                break;
            case TK_RX_START:
                if (pos == start && (pos == 0 || t.hard(pos))) ++pc;
                else fail = true;
                break;
            case TK_RX_ATOM_BEGIN:
                if (sp == TK_RX_STACK) return TK_RX_OVERFLOW;
                fk[sp] = (uint32_t)F_ATOM << 24;
                fp[sp] = fa_[sp] = 0;
                ++sp;
                ++pc;
                break;
            case TK_RX_ATOM_END:  // the group matched: its alternatives are forgotten
                while (sp > 0 && (fk[--sp] >> 24) != F_ATOM) {}
                ++pc;
                break;
            case TK_RX_LOOK_BEGIN:
                if (sp == TK_RX_STACK) return TK_RX_OVERFLOW;
                fk[sp] = (uint32_t)F_LOOK << 24 | I.b;
                fp[sp] = pos;
                fa_[sp] = I.a;
                ++sp;
                ++pc;
                break;
            case TK_RX_LOOK_END: {  // the look-ahead's body matched
                while (sp > 0 && (fk[--sp] >> 24) != F_LOOK) {}
                pos = fp[sp];
                if (fa_[sp]) fail = true;  // negative look-ahead
                else ++pc;
            } break;
            case TK_RX_POP:
                if (sp > 0) --sp;
                ++pc;
                break;
            case TK_RX_WORDB:
            case TK_RX_PREV: {
                // the d-th char before the position, if the haystack has one (it begins at a hard start; the match itself may have begun later)
                const uint32_t dist = (I.op & 0xFFu) == TK_RX_PREV ? I.c : 1u;
                uint32_t q = pos, prev = 0;
                bool has_prev = true;
                for (uint32_t d = 0; d < dist && has_prev; ++d) {
                    has_prev = q > 0 && !(q <= start && t.hard(q));  // (a hard position at or before the match's start is where the haystack begins)
                    if (!has_prev) break;
                    uint32_t r = q - 1, len;
                    for (int k = 0; k < 3 && r > 0 && (t.byte(r) & 0xC0u) == 0x80u && !(r <= start && t.hard(r)); ++k) --r;
                    prev = tk_rx_decode(t, r, &len);
                    if (r + len != q) {  // (malformed UTF-8: the byte before stands for itself)
                        prev = 0xFFFDu;
                        r = q - 1;
                    }
                    q = r;
                }
                bool ok;
                if ((I.op & 0xFFu) == TK_RX_PREV) {
                    ok = has_prev && tk_rx_in_set(P, I.a, prev);
                } else {
                    uint32_t len;
                    const bool wb = has_prev && tk_rx_is_word(P, prev), wa = !at_end(pos) && tk_rx_is_word(P, tk_rx_decode(t, pos, &len));
                    ok = (wb != wa) != (bool)I.a;
                }
                if (ok) ++pc;
                else fail = true;
            } break;
            default: fail = true; break;
        }
        while (fail) {  // backtrack
            if (sp == 0) {
                TK_RX_ON_DONE(steps);
                return TK_RX_FAILED;
            }
            if (++steps > (3u << 30)) return TK_RX_LIMIT;  // (a pop is a step; the check proper is at the next instruction)
            --sp;
            const uint32_t kind = fk[sp] >> 24, tgt = fk[sp] & 0xFFFFFFu;
            if (kind == F_ALT) {
                pc = tgt;
                pos = fp[sp];
                fail = false;
            } else if (kind == F_RANGE) {  // give back one char of a greedy run: [fp, fa] is the range of possible ends
                uint32_t cur = fa_[sp] - 1;
                while (cur > fp[sp] && (t.byte(cur) & 0xC0u) == 0x80u) --cur;
                if (cur > fp[sp]) {
                    fa_[sp] = cur;
                    ++sp;
                }
                pos = cur;
                pc = tgt;
                fail = false;
            } else if (kind == F_LAZY) {  // take one more char of a lazy run
                const TkRxIns R = P.ins[tgt];
                uint32_t p = fp[sp], c = fa_[sp], len;
                if (c < R.c && !at_end(p) && tk_rx_in_set(P, R.a, tk_rx_decode(t, p, &len))) {
                    p += len;
                    ++c;
                    if (c < R.c) {
                        fp[sp] = p;
                        fa_[sp] = c;
                        ++sp;
                    }
                    pos = p;
                    pc = tgt + 1;
                    fail = false;
                }
            } else if (kind == F_LOOK) {  // the look-ahead's body cannot match
                if (fa_[sp]) {             // ... which is what a negative one asks for
                    pos = fp[sp];
                    pc = tgt;
                    fail = false;
                }
            }  // F_ATOM: the group failed as a whole
        }
    }
}

// DFA: 0 = the program; 1 = the table form (P.dfa_trans is there); 2 = the table of a pattern that looks behind (P.dfa_flags bit 0)
#define TK_RX_M_PROGRAM 0
#define TK_RX_M_DFA 1
#define TK_RX_M_DFA_PREV 2
template <int DFA, class A>
TK_HD uint32_t tk_rx_match_sel(const TkRxProg& P, A& t, uint32_t start) {
    if constexpr (DFA == TK_RX_M_DFA_PREV) return tk_rx_match_dfa<true>(P, t, start);
    else if constexpr (DFA == TK_RX_M_DFA) return tk_rx_match_dfa<false>(P, t, start);
    else return tk_rx_match(P, t, start);
}
