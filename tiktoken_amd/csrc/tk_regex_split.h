// Piece starts of a chunk under a compiled pat_str (tk_regex.h), found in parallel and then proven -- the work of one lane of each of the
// two kernels of tk_regex_kernels.h, written for host and device so that the CPU tests can run the very same code lane by lane.
//
// The chain of piece starts of a document is sequential: the next start is where the match that begins at this one ends (reference:
// find_iter, src/lib.rs:365).  But the end of the match that starts at p depends only on the text from p on (no look-behind), so:
//   1. tk_rx_speculate_lane: the text is cut into segments of 256 or 1024 bytes; lane k starts at the first char of segment k AS IF it
//      were a piece start and follows the chain to the end of its segment, noting every start in the bitmap `spec` and the position at which
//      it leaves the segment in `xexit[k]`.  Chains that start at different places run together after a few pieces, so most of these
//      guesses are right from some point on -- but nothing here is trusted yet.
//   2. tk_rx_resolve_lane: one lane per document walks the TRUE chain from the document start.  Wherever the true start p it has reached
//      is on segment k's speculative chain (its bit is set in `spec`), everything that chain found from p on is true as well: the lane takes
//      the rest of the segment's bits and jumps to xexit[k] -- a few memory operations per KiB instead of a match per piece.  Where p
//      is not on the chain it runs the matcher itself until the chains meet.  The result is exact whatever the guesses were.
//   1b. tk_rx_link_lane: lane k walks from xexit[k - 1] -- where the true chain enters segment k if the guess for k - 1 was right -- until
//      it meets segment k's chain; with these links the resolving pass runs the matcher only where a guess was wrong, and a group of
//      lanes can prove 64 consecutive segments of one document at once (tk_rx_plan / tk_rx_emit).
// A speculative match never reads more than TK_RX_AHEAD bytes beyond its segment (a megabyte of one letter is one piece: every lane inside
// it would scan to its end); a lane that would have to at its FIRST evaluation stops (it is probably inside that piece), one that meets
// such a piece later along its chain evaluates it in full -- once, in parallel with everybody else -- and the resolving lane, which really
// is at a piece start, evaluates whatever is left.
//
// Special tokens (encode() with allowed_special): a haystack ends where a special token starts (`hard` bit, as at a document start);
// the token itself is one step of the chain.  A position where the pattern does not match: the reference's find_iter goes on to the next
// match and the text in between yields no token (src/lib.rs:365,405).  Here such a char is a step of the chain as well -- a GAP piece of
// one char, marked in a second bitmap (`sgap` beside `spec`, `ggap` beside the true starts) -- and the pipeline behind the split gives a
// gap piece no token (TK_RES_GAP, tk_fused.h).
#pragma once
#include "tk_regex.h"

// bytes per speculative segment = 1 << seg_shift (a multiple of 32: a segment owns its words of the bitmap).  Measured with the o200k
// pat_str on 256 MiB of the bench corpus (profiles/r03_generic_engine.txt), speculative pass: 1 KiB segments 24.2 ms, 512 B 20.8, 256 B
// 17.8 / 14.9 (with the link pass), 128 B 13.6, 64 B 13.5 -- more lanes beat the pieces that are matched twice around the segment
// boundaries; 128 bytes for every chunk (the CPU tests also run the LARGE form).
#define TK_RX_SEG_SHIFT_SMALL 7u
#define TK_RX_SEG_SHIFT_LARGE 10u
#define TK_RX_SEG_SMALL_BELOW (4ull << 30)  // chunk bytes below which the SMALL segments are used: always
#define TK_RX_AHEAD 16384u  // bytes a speculative match may look beyond its segment: the program (its lanes evaluate longer pieces they come to in full)
#define TK_RX_AHEAD_DFA 2048u  // ... the DFA forms: a longer piece is left to the resolving wavefront, which takes it a KiB per step (tk_rx_match_dfa_coop)
#define TK_RX_UNKNOWN 0xFFFFFFFFu
#define TK_RX_ERR_GAP 4u       // bits of the chunk's error word
#define TK_RX_ERR_STACK 8u
#define TK_RX_ERR_LIMIT 16u
#ifndef TK_RX_ON_MATCH
#define TK_RX_ON_MATCH()  // (the CPU tests count the matcher's runs here)
#endif

struct TkRxText {
    const uint8_t* text;
    uint32_t n;
    const uint32_t* brk;  // hard starts: documents, special-token edges
    const uint32_t* ss;   // special-token starts (null: none)
    const uint32_t* si;   // special-token interiors
    uint32_t limit;       // a speculative match sees the end of the text here ...
    bool hit;             // ... and says so
    // the text and the bitmap word read last: a lane walks its text byte by byte, one load per sixteen bytes (device; the lanes of a
    // wavefront read sixty-four different cache lines per load instruction: fewer, wider loads) / four bytes (host) / thirty-two positions
    uint32_t tw_at = 0xFFFFFFFFu, tw = 0, bw_at = 0xFFFFFFFFu, bw = 0;
    uint32_t ahead = TK_RX_AHEAD;  // bytes a speculative match may look beyond its segment
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t tq1 = 0, tq2 = 0, tq3 = 0;  // (with tw: the sixteen bytes at 16 * tw_at)
    TK_HD uint32_t byte(uint32_t p) {
        const uint32_t i = p >> 4;
        if (i != tw_at) {
            tw_at = i;
            const uint4 q = *(const uint4*)(text + 16u * (size_t)i);  // (the chunk's text is 16-byte aligned and readable 64 bytes past n)
            tw = q.x;
            tq1 = q.y;
            tq2 = q.z;
            tq3 = q.w;
        }
        const uint32_t lo = (p & 8u) ? tq2 : tw, hi = (p & 8u) ? tq3 : tq1;
        return (((p & 4u) ? hi : lo) >> (8u * (p & 3u))) & 0xFFu;
    }
#else
    TK_HD uint32_t byte(uint32_t p) {
        const uint32_t i = p >> 2;
        if (i != tw_at) {
            tw_at = i;
            __builtin_memcpy(&tw, text + 4u * (size_t)i, 4);
        }
        return (tw >> (8u * (p & 3u))) & 0xFFu;
    }
#endif
    // (for tk_rx_run_lane: an aligned word, an aligned 16-byte block, the hard bits of the 16 positions of a block -- not cached)
    TK_HD uint32_t word(uint32_t p4) const {
        uint32_t v;
#if defined(__HIP_DEVICE_COMPILE__)
        v = *(const uint32_t*)(text + (size_t)p4);
#else
        __builtin_memcpy(&v, text + (size_t)p4, 4);
#endif
        return v;
    }
    TK_HD void block16(uint32_t p16, uint32_t* w) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint4 q = *(const uint4*)(text + (size_t)p16);
        w[0] = q.x;
        w[1] = q.y;
        w[2] = q.z;
        w[3] = q.w;
#else
        __builtin_memcpy(w, text + (size_t)p16, 16);
#endif
    }
    TK_HD uint32_t hard16(uint32_t p16) const { return (brk[p16 >> 5] >> (p16 & 31u)) & 0xFFFFu; }
    TK_HD bool hard(uint32_t p) {
        if (p >= limit) {
            hit = true;
            return true;
        }
        const uint32_t i = p >> 5;
        if (i != bw_at) {
            bw_at = i;
            bw = brk[i];
        }
        return (bw >> (p & 31u)) & 1u;
    }
    TK_HD bool special(uint32_t p) const { return ss && ((ss[p >> 5] >> (p & 31u)) & 1u); }
    TK_HD bool inside_special(uint32_t p) const { return si && ((si[p >> 5] >> (p & 31u)) & 1u); }
};

// the start that follows the piece (or special token, or gap char) that starts at p; or an error code (TK_RX_IS_ERROR: stack, budget).
// DFA (here and below): 0 = the program, 1 = the pattern's table form (tk_rx_match_dfa), 2 = the table of a pattern that looks behind.
// (match(p): the end of the match that starts at p -- tk_rx_match_sel, or a group of lanes together: tk_rx_match_dfa_coop)
template <class Match>
TK_HD uint32_t tk_rx_next_with(const TkRxProg& P, TkRxText& t, uint32_t p, bool* gap, Match&& match) {
    *gap = false;
    if (t.special(p)) {
        uint32_t q = p + 1;
        while (q < t.n && !t.hard(q)) ++q;
        return q;
    }
    TK_RX_ON_MATCH();
    const uint32_t e = match(p);
    if (e != TK_RX_FAILED) return e;
    // no match starts here: the char is skipped (find_iter tries the next position)
    *gap = true;
    uint32_t q = p + 1;
    while (q < t.n && !t.hard(q) && (t.byte(q) & 0xC0u) == 0x80u) ++q;
    return q;
}

template <int DFA = 0>
TK_HD uint32_t tk_rx_next(const TkRxProg& P, TkRxText& t, uint32_t p, bool* gap) {
    return tk_rx_next_with(P, t, p, gap, [&](uint32_t at) { return tk_rx_match_sel<DFA>(P, t, at); });
}

// bit p of a bitmap
TK_HD bool tk_rx_bit(const uint32_t* bm, uint32_t p) { return (bm[p >> 5] >> (p & 31u)) & 1u; }

// the chain of segment [.., end) from its start p on (first: the segment's first start), as described above; returns the segment's exit
template <int DFA = 0>
TK_HD uint32_t tk_rx_speculate_chain(const TkRxProg& P, TkRxText& t, uint32_t p, uint32_t first, uint32_t end, uint32_t limit, uint32_t* spec, uint32_t* sgap) {
    for (;;) {
        spec[p >> 5] |= 1u << (p & 31u);  // (the words of a segment belong to its lane)
        bool gap;
        uint32_t q = tk_rx_next<DFA>(P, t, p, &gap);
        if (!DFA && t.hit && p != first && !TK_RX_IS_ERROR(q)) {
            // A long piece (it reaches TK_RX_AHEAD bytes beyond the segment) that this lane has come to along its chain: very likely a
            // true start, and nobody else will evaluate it in parallel -- the lanes of the segments inside the piece stop at their FIRST
            // evaluation (below), so the text of a long piece is scanned once here instead of once by the resolving lane of its document.
            // (The program only: with the pattern's DFA the resolving pass takes a long piece a KiB per step, the lanes of its
            // wavefront together -- tk_rx_match_dfa_coop -- and a lane on its own has no business with it.)
            t.limit = 0xFFFFFFFFu;
            t.hit = false;
            q = tk_rx_next<DFA>(P, t, p, &gap);
            t.limit = limit;
        }
        if (TK_RX_IS_ERROR(q) || t.hit) return TK_RX_UNKNOWN;
        if (gap) sgap[p >> 5] |= 1u << (p & 31u);
        if (q >= end) return q;
        p = q;
    }
}

template <int DFA = 0>
TK_HD void tk_rx_speculate_lane(const TkRxProg& P, TkRxText t, uint32_t k, uint32_t seg_shift, uint32_t* spec, uint32_t* sgap, uint32_t* xexit) {
    const uint64_t a64 = (uint64_t)k << seg_shift;
    if (a64 >= t.n) return;
    const uint32_t seg = 1u << seg_shift;
    const uint32_t a = (uint32_t)a64, end = t.n - a > seg ? a + seg : t.n;
    const uint32_t limit = t.n - end > t.ahead ? end + t.ahead : t.n;
    t.limit = limit;
    t.hit = false;
    uint32_t p = a;
    while (p < end && ((t.byte(p) & 0xC0u) == 0x80u || t.inside_special(p))) ++p;
    xexit[k] = p < end ? tk_rx_speculate_chain<DFA>(P, t, p, p, end, limit, spec, sgap) : TK_RX_UNKNOWN;
}

// The same with the pattern's DFA, as ONE loop over the chars of the segment: where a piece ends the lane notes its start and goes on with
// the next one in the same iteration scheme, so the lanes of a wavefront -- whose pieces begin and end at different places -- all do useful
// work in every iteration (in the form above a wavefront takes as long for every piece as its longest piece, for every segment as many
// pieces as its busiest lane has).  The bits of a bitmap word are collected in a register and stored once.  What is rare leaves the loop
// and goes on in tk_rx_speculate_chain from the piece at hand: a special token, a match that looks TK_RX_AHEAD bytes beyond the segment.
// Same bitmaps and exits as tk_rx_speculate_lane<true> (the CPU tests compare them bit for bit).
template <bool PREV = false>
TK_HD void tk_rx_speculate_lane_flat(const TkRxProg& P, TkRxText t, uint32_t k, uint32_t seg_shift, uint32_t* spec, uint32_t* sgap, uint32_t* xexit) {
    const uint64_t a64 = (uint64_t)k << seg_shift;
    if (a64 >= t.n) return;
    const uint32_t seg = 1u << seg_shift;
    const uint32_t a = (uint32_t)a64, end = t.n - a > seg ? a + seg : t.n;
    const uint32_t limit = t.n - end > t.ahead ? end + t.ahead : t.n;
    t.limit = limit;
    t.hit = false;
    uint32_t p = a;
    while (p < end && ((t.byte(p) & 0xC0u) == 0x80u || t.inside_special(p))) ++p;
    if (p >= end) {
        xexit[k] = TK_RX_UNKNOWN;
        return;
    }
    const uint32_t first = p, ncls = P.dfa_ncls;
    uint32_t sw = p >> 5, sbits = 0u, gbits = 0u;  // the word of `spec` / `sgap` being filled
    auto flush = [&]() {
        if (sbits) spec[sw] |= sbits;
        if (gbits) sgap[sw] |= gbits;
        sbits = gbits = 0u;
    };
    auto note_start = [&](uint32_t q) {
        if ((q >> 5) != sw) {
            flush();
            sw = q >> 5;
        }
        sbits |= 1u << (q & 31u);
    };
    note_start(p);
    uint32_t x = TK_RX_UNKNOWN;
    bool slow = t.special(p);
    uint32_t pos = p, last = TK_RX_FAILED, state = tk_rx_dfa_start<PREV>(P, t, p);
    while (!slow) {
        uint32_t cls = 0u, len = 0u;  // (the end of the haystack)
        if (pos < t.n) {
            if (pos >= limit) {  // the match looks too far ahead: the general form decides what to do with this piece
                slow = true;
                break;
            }
            if (!(pos > p && t.hard(pos))) {
                const uint32_t b0 = t.byte(pos);
                if (b0 < 0x80u) {
                    cls = tk_rx_ascii_cls(P, b0);
                    len = 1u;
                } else {
                    cls = tk_rx_dfa_cls(P, tk_rx_decode(t, pos, &len));
                }
            }
        }
        const uint32_t e = P.dfa_trans[state * ncls + cls];
        if (e & 0x8000u) last = pos;
        state = e & 0x7FFFu;
        if (state != 0u) {
            pos += len;
            continue;
        }
        // the piece that starts at p is finished: it ends at `last`, or p is a char the pattern does not match (a gap piece of one char)
        TK_RX_ON_MATCH();
        TK_RX_ON_DONE(pos - p + 1u);
        uint32_t q = last;
        if (last == TK_RX_FAILED) {
            q = p + 1u;
            while (q < t.n && !t.hard(q) && (t.byte(q) & 0xC0u) == 0x80u) ++q;
            if (t.hit) {  // (the char ends at the look-ahead limit: cannot happen within a segment, kept for the general form to decide)
                t.hit = false;
                slow = true;
                break;
            }
            gbits |= 1u << (p & 31u);
        }
        if (q >= end) {
            x = q;
            break;
        }
        p = q;
        note_start(p);
        if (t.special(p)) {
            slow = true;
            break;
        }
        pos = p;
        last = TK_RX_FAILED;
        if constexpr (PREV) state = tk_rx_dfa_start<true>(P, t, p);
        else state = t.hard(p) ? 1u : 2u;  // (p > 0 here)
    }
    flush();
    if (slow) x = tk_rx_speculate_chain<PREV ? TK_RX_M_DFA_PREV : TK_RX_M_DFA>(P, t, p, first, end, limit, spec, sgap);
    xexit[k] = x;
}

// ---- The speculative pass over STAGED text (round 5: tk_k_rx_speculate_staged, tk_regex_kernels.h) ----------------------------------
// The one-loop lane above reads its text, its bits of `brk` and its words of `spec` from global memory as it walks: sixty-four different
// cache lines per load instruction, and a wait in nearly every iteration of the wavefront's loop.  Here a workgroup first turns the 32 KiB of
// its 256 segments (plus TK_RX_STAGE_HALO bytes behind them) into one CODE per byte, in LDS -- whole 16-byte blocks per lane, every load
// coalesced, the UTF-8 decoding and the class look-ups done once per char instead of once per visit -- and the lanes then walk codes:
//   0 .. TK_RX_CODE_MAX_CLS   the DFA class of the char that starts at this byte (0: the end of the text)
//   | TK_RX_CODE_HARD         ... and a hard start (a document begins, the edge of a special token)
//   TK_RX_CODE_CONT           a continuation byte of a well-formed char (the walk steps over it)
//   TK_RX_CODE_ESC            anything else: a special token, bytes that are not well-formed UTF-8, a hard start inside a char
// A lane that meets TK_RX_CODE_ESC, leaves the staged stretch or looks further ahead than the one-loop lane may (TkRxText::ahead) gives
// its segment to that lane -- whatever is unusual is decided by the code that has been compared with the matcher for four rounds, and the
// new loop only ever does what that code does on plain text.  It keeps the bits of its four words of `spec` / `sgap` in registers and
// stores them once.  The CPU tests run both forms side by side on every text they split and compare
// bitmaps and exits bit for bit.
#define TK_RX_CODE_CONT 0x7Fu
#define TK_RX_CODE_ESC 0x7Eu
#define TK_RX_CODE_HARD 0x80u
#define TK_RX_CODE_MAX_CLS 0x7Du
#ifndef TK_RX_STAGE_SEGS
#define TK_RX_STAGE_SEGS 256u   // segments (of 1 << TK_RX_SEG_SHIFT_SMALL bytes) per workgroup = its lanes
#endif
#define TK_RX_STAGE_HALO 256u   // bytes staged behind them (a multiple of the segment)
#define TK_RX_STAGE_BYTES ((TK_RX_STAGE_SEGS << TK_RX_SEG_SHIFT_SMALL) + TK_RX_STAGE_HALO)

// Where the code of the byte at offset o of the staged stretch lives: the lanes of a wavefront stand at about the same offset of their
// segments, 128 bytes apart -- one LDS bank -- so the words of segment s are permuted by s (an exclusive-or: three instructions).
TK_HD uint32_t tk_rx_code_addr(uint32_t o) { return o ^ (((o >> 7) & 31u) << 2); }

struct TkRxCodes {
    const uint8_t* c;  // codes of the positions [r0, r0 + len)
    uint32_t r0, len;
    TK_HD uint32_t at(uint32_t pos) const {
        const uint32_t o = pos - r0;
        return o < len ? (uint32_t)c[tk_rx_code_addr(o)] : TK_RX_CODE_ESC;
    }
};
// the four words of codes of the 16-byte block at offset o (a multiple of 16)
TK_HD void tk_rx_codes_store(uint8_t* c, uint32_t o, const uint32_t w[4]) {
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) *(uint32_t*)(c + tk_rx_code_addr(o + 4u * i)) = w[i];  // (whole words move: the permutation leaves the low two bits alone)
}

// bits of a bitmap for the 24 positions blk - 4 .. blk + 19 (blk a multiple of 16; the bitmaps have two words of slack)
TK_HD uint32_t tk_rx_bits24(const uint32_t* bm, uint32_t blk) {
    if (blk == 0u) return (bm[0] << 4) & 0xFFFFFFu;
    const uint32_t q = blk - 4u;
    const uint64_t v = (uint64_t)bm[q >> 5] | ((uint64_t)bm[(q >> 5) + 1u] << 32);
    return (uint32_t)(v >> (q & 31u)) & 0xFFFFFFu;
}
// bits 7, 15, 23, 31 of m -> bits 0 .. 3
TK_HD uint32_t tk_rx_pack4(uint32_t m) { return ((((m >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; }

// the codes of the 16-byte block at blk (a multiple of 16), four to a word
template <class A>
TK_HD void tk_rx_codes16(const TkRxProg& P, const A& t, uint32_t blk, uint32_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0u;
    if (blk >= t.n) return;
    uint32_t w[6];  // bytes blk - 4 .. blk + 19: the chars that reach into the block or out of it
    w[0] = blk ? t.word(blk - 4u) : 0u;
    t.block16(blk, w + 1);
    w[5] = t.word(blk + 16u);  // (the text is readable 64 bytes past n)
    const uint32_t hard = tk_rx_bits24(t.brk, blk);
    const uint32_t spc = t.ss ? (tk_rx_bits24(t.ss, blk) | tk_rx_bits24(t.si, blk)) : 0u;
    const uint32_t left = t.n - blk;  // (> 0)
    uint32_t valid = left >= 20u ? 0xFFFFFFu : ((1u << (left + 4u)) - 1u);  // window positions that are text
    if (blk == 0u) valid &= ~0xFu;
    // byte kinds over the window, a bit per position
    uint32_t isc = 0u, l2 = 0u, l3 = 0u, l4 = 0u;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t m80 = w[i] & 0x80808080u, m40 = (w[i] << 1) & 0x80808080u, m20 = (w[i] << 2) & 0x80808080u, m10 = (w[i] << 3) & 0x80808080u;
        const uint32_t lead = m80 & m40;
        isc |= tk_rx_pack4(m80 & ~m40) << (4 * i);
        l2 |= tk_rx_pack4(lead & ~m20) << (4 * i);
        l3 |= tk_rx_pack4(lead & m20 & ~m10) << (4 * i);
        l4 |= tk_rx_pack4(lead & m20 & m10) << (4 * i);
    }
    isc &= valid;
    l2 &= valid;
    l3 &= valid;
    l4 &= valid;
    // a lead is well formed when the bytes it needs are continuation bytes of the text without a hard start or a special token on them;
    // a continuation byte is part of a char when such a lead covers it
    const uint32_t clean = isc & ~(hard | spc);
    const uint32_t c1 = clean >> 1, c2 = clean >> 2, c3 = clean >> 3;
    const uint32_t ok2 = l2 & c1, ok3 = l3 & c1 & c2, ok4 = l4 & c1 & c2 & c3;
    const uint32_t cov = (ok2 << 1) | (ok3 << 1) | (ok3 << 2) | (ok4 << 1) | (ok4 << 2) | (ok4 << 3);
    const uint32_t esc = ((isc & ~cov) | ((l2 | l3 | l4) & ~(ok2 | ok3 | ok4)) | spc) & valid;
#define TK_RX_B(j) ((w[(j) >> 2] >> (8u * ((j) & 3u))) & 0xFFu)
    // the ASCII classes of all sixteen bytes first -- sixteen independent look-ups in flight, one wait -- then chars beyond ASCII, where the block has any
    uint32_t cls[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cls[i] = tk_rx_ascii_cls(P, TK_RX_B(i + 4) & 0x7Fu);
    const uint32_t leads = ((ok2 | ok3 | ok4) >> 4) & 0xFFFFu;
    if (leads) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i + 4;
            if ((leads >> i) & 1u) {
                const uint32_t b = TK_RX_B(j);
                const uint32_t need = ((l2 >> j) & 1u) ? 2u : (((l3 >> j) & 1u) ? 3u : 4u);
                uint32_t cp = ((b & (0x7Fu >> need)) << 6) | (TK_RX_B(j + 1) & 0x3Fu);
                if (need >= 3u) cp = (cp << 6) | (TK_RX_B(j + 2) & 0x3Fu);
                if (need == 4u) cp = (cp << 6) | (TK_RX_B(j + 3) & 0x3Fu);
                cls[i] = tk_rx_dfa_cls(P, cp);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = i + 4;
        uint32_t code = cls[i] | (((hard >> j) & 1u) ? TK_RX_CODE_HARD : 0u);
        if ((isc >> j) & 1u) code = TK_RX_CODE_CONT;
        if ((esc >> j) & 1u) code = TK_RX_CODE_ESC;
        if (!((valid >> j) & 1u)) code = 0u;  // (at or behind the end of the text)
        out[i >> 2] |= code << (8u * ((uint32_t)i & 3u));
    }
#undef TK_RX_B
}

// Lane k over the codes: the chain of segment k as tk_rx_speculate_lane_flat<false> finds it -- its bits of the segment's four words of
// `spec` / `sgap` in sb / gb, its exit in *xout -- or false: the segment is that lane's (nothing has been written).
// trans_pm: the transition table with the ROW OFFSET of the next state in place of its number (tk_rx_trans_premultiplied: the look-up is
// trans_pm[row + class], no multiplication in the chain from one char to the next).
TK_HD uint16_t tk_rx_trans_premultiplied(uint16_t e, uint32_t ncls) { return (uint16_t)((e & 0x8000u) | ((e & 0x7FFFu) * ncls)); }  // (states x classes < 32768: tk_rx_staged_fits)
TK_HD bool tk_rx_speculate_lane_codes(const uint16_t* trans_pm, uint32_t ncls, const TkRxCodes& C, uint32_t n, uint32_t ahead, uint32_t k, uint32_t sb[4], uint32_t gb[4], uint32_t* xout) {
    const uint32_t a = k << TK_RX_SEG_SHIFT_SMALL, seg = 1u << TK_RX_SEG_SHIFT_SMALL;
    const uint32_t end = n - a > seg ? a + seg : n;
    const uint32_t limit = n - end > ahead ? end + ahead : n;
    sb[0] = sb[1] = sb[2] = sb[3] = gb[0] = gb[1] = gb[2] = gb[3] = 0u;
    *xout = TK_RX_UNKNOWN;
    auto note = [&](uint32_t* v, uint32_t q) {
        const uint32_t wq = (q >> 5) & 3u, m = 1u << (q & 31u);  // (the segment starts at a multiple of 128)
        v[0] |= wq == 0u ? m : 0u;
        v[1] |= wq == 1u ? m : 0u;
        v[2] |= wq == 2u ? m : 0u;
        v[3] |= wq == 3u ? m : 0u;
    };
    uint32_t p = a, code;
    for (;;) {  // the first char of the segment
        if (p >= end) return true;
        code = C.at(p);
        if (code == TK_RX_CODE_ESC) return false;
        if (code != TK_RX_CODE_CONT) break;
        ++p;
    }
    note(sb, p);
    // ONE loop, one code per iteration: a continuation byte is a step of its own, the first char of a piece sets the state
    uint32_t pos = p, last = TK_RX_FAILED, state = 0u;
    bool ok = true;
    // where the lane gives up: the end of the staged stretch, or the limit of the look-ahead where the text goes on behind it (pos moves a
    // byte at a time and never passes the end of the text: it meets `limit` before it could pass it)
    uint32_t stop = C.len;
    if (limit < n && limit - C.r0 < stop) stop = limit - C.r0;
    for (;;) {
        const uint32_t o = pos - C.r0;
        code = C.c[tk_rx_code_addr(o < C.len ? o : C.len - 1u)];
        if (o >= stop) code = TK_RX_CODE_ESC;
        if (code == TK_RX_CODE_ESC) {  // (the general form decides what to do with this piece)
            ok = false;
            break;
        }
        if (code == TK_RX_CODE_CONT) {  // (never at pos == p: a piece starts at a char)
            ++pos;
            continue;
        }
        const bool hard = (code & TK_RX_CODE_HARD) != 0u;
        if (pos == p) state = (p == 0u || hard) ? ncls : 2u * ncls;  // (state 1 or 2, as a row offset; a char has a class > 0 and is consumed: pos == p at the first step of a piece only)
        const uint32_t c = (hard && pos > p) ? 0u : (code & 0x7Fu);  // (class 0: the end of the haystack)
        const uint32_t e = trans_pm[state + c];
        if (e & 0x8000u) last = pos;
        state = e & 0x7FFFu;
        if (state != 0u) {
            pos += c != 0u ? 1u : 0u;  // (nothing consumes the end of the haystack: the state dies there after a step or two)
            continue;
        }
        // the piece that starts at p is finished: it ends at `last`, or p is a char the pattern does not match (a gap piece of one char)
        uint32_t q = last;
        if (last == TK_RX_FAILED) {
            q = p + 1u;
            uint32_t cq;
            while ((cq = C.at(q)) == TK_RX_CODE_CONT) ++q;
            if (cq == TK_RX_CODE_ESC) {
                ok = false;
                break;
            }
            note(gb, p);
        }
        if (q >= end) {
            *xout = q;
            break;
        }
        p = q;
        note(sb, p);
        pos = p;
        last = TK_RX_FAILED;
    }
    return ok;
}

// Links between consecutive segments.  The true chain enters segment k where segment k - 1 was left -- at xexit[k - 1], if that guess was
// right -- and that position is usually NOT on segment k's own speculative chain (which started at the segment's first char): the chains
// meet a few pieces later.  Lane k walks from xexit[k - 1] until it stands on a start of segment k's chain, noting its steps in `lnk`
// (gap chars in `lgap`; both in the segment's own words) and the meeting point in lmerge[k]; or it leaves the segment without meeting the
// chain (lmerge[k] = end of the segment, lexit[k] = where to).  With the links the resolving pass runs the matcher only where a guess was
// wrong: entering segment k at xexit[k - 1] it takes the link's steps, then the chain's from the meeting point on, and jumps to xexit[k].
#define TK_RX_NOLINK 0xFFFFFFFFu
template <int DFA = 0>
TK_HD void tk_rx_link_lane(const TkRxProg& P, TkRxText t, uint32_t k, uint32_t seg_shift, const uint32_t* spec, const uint32_t* xexit,
                           uint32_t* lnk, uint32_t* lgap, uint32_t* lmerge, uint32_t* lexit) {
    const uint64_t a64 = (uint64_t)k << seg_shift;
    if (a64 >= t.n) return;
    const uint32_t seg = 1u << seg_shift;
    const uint32_t a = (uint32_t)a64, end = t.n - a > seg ? a + seg : t.n;
    uint32_t m = TK_RX_NOLINK, x = TK_RX_UNKNOWN;
    const uint32_t e = k ? xexit[k - 1] : TK_RX_UNKNOWN;
    if (e != TK_RX_UNKNOWN && e >= a && e < end) {
        if (tk_rx_bit(spec, e)) {
            m = e;  // on the chain as it is
        } else {
            t.limit = t.n - end > t.ahead ? end + t.ahead : t.n;
            t.hit = false;
            uint32_t p = e;
            for (;;) {
                lnk[p >> 5] |= 1u << (p & 31u);
                bool gap;
                const uint32_t q = tk_rx_next<DFA>(P, t, p, &gap);
                if (TK_RX_IS_ERROR(q) || t.hit) break;  // (no link: the resolving lane evaluates this stretch itself)
                if (gap) lgap[p >> 5] |= 1u << (p & 31u);
                if (q >= end) {
                    m = end;
                    x = q;
                    break;
                }
                if (tk_rx_bit(spec, q)) {
                    m = q;
                    break;
                }
                p = q;
            }
        }
    }
    lmerge[k] = m;
    lexit[k] = x;
}

struct TkRxMaps {  // what the speculative pass and the link pass have left (lnk == nullptr: no link pass; spec == nullptr: no speculation)
    const uint32_t *spec, *sgap, *xexit, *lnk, *lgap, *lmerge, *lexit;
    uint32_t seg_shift;
};

// the set bits of bm (gaps: gm) in [from, to), through `orbits`; returns the last position taken, or `none`
template <class Or>
TK_HD uint32_t tk_rx_take(const uint32_t* bm, const uint32_t* gm, uint32_t from, uint32_t to, uint32_t none, Or&& orbits) {
    uint32_t last = none;
    if (from >= to) return last;
    for (uint32_t w = from >> 5; w <= (to - 1u) >> 5; ++w) {
        uint32_t bits = bm[w];
        if (w == (from >> 5)) bits &= ~0u << (from & 31u);
        if (w == ((to - 1u) >> 5) && (to & 31u)) bits &= (1u << (to & 31u)) - 1u;
        if (bits) {
            last = w * 32u + 31u - (uint32_t)__builtin_clz(bits);
            orbits(w, bits, gm[w] & bits);  // (a start whose evaluation a guess broke off has no gap bit: it is matched again)
        }
    }
    return last;
}

// What segment k holds for a chain that enters it at `entry` (a true start, or assumed to be one), for the document that ends at e:
//   ok    : the maps answer -- entry is on the segment's chain, or it is xexit[k - 1] and the link pass has walked from there;
//   m     : from where on the segment's own chain is taken (>= seg_end: nowhere), link: the steps of `lnk` before it are taken too;
//   exit  : the first start behind the segment (>= e: the document is finished); TK_RX_UNKNOWN: the chain's guess broke off (a piece that
//           looks too far ahead) -- the caller goes on from the last start taken.
struct TkRxPlan {
    bool ok, link;
    uint32_t m, seg_end, exit;
};
TK_HD TkRxPlan tk_rx_plan(const TkRxMaps& M, uint32_t n, uint32_t k, uint32_t entry, uint32_t e) {
    TkRxPlan R{false, false, 0, 0, TK_RX_UNKNOWN};
    const uint64_t a64 = (uint64_t)k << M.seg_shift, end64 = a64 + (1ull << M.seg_shift);
    if (!M.spec || a64 >= n) return R;
    const uint32_t a = (uint32_t)a64, end = end64 < n ? (uint32_t)end64 : n;
    R.seg_end = end < e ? end : e;
    if (entry < a || entry >= R.seg_end) return R;
    if (tk_rx_bit(M.spec, entry)) {
        R.m = entry;
    } else if (M.lnk && k && M.xexit[k - 1] == entry && M.lmerge[k] != TK_RX_NOLINK) {
        R.m = M.lmerge[k];
        R.link = true;
    } else {
        return R;
    }
    R.ok = true;
    if (R.m >= R.seg_end) R.exit = (R.link && R.m == end) ? M.lexit[k] : R.m;  // (left without meeting the chain; or met it behind the document's end)
    else R.exit = M.xexit[k];
    return R;
}
// the starts of the plan, through `orbits`; returns the last start taken
template <class Or>
TK_HD uint32_t tk_rx_emit(const TkRxMaps& M, const TkRxPlan& R, uint32_t entry, Or&& orbits) {
    uint32_t last = entry;
    if (R.link) last = tk_rx_take(M.lnk, M.lgap, entry, R.m < R.seg_end ? R.m : R.seg_end, last, orbits);
    if (R.m < R.seg_end) last = tk_rx_take(M.spec, M.sgap, R.m, R.seg_end, last, orbits);
    return last;
}

// One step of the true chain of the document [.., e) from the true start p: the whole rest of p's segment when the maps answer, one
// match otherwise.  Returns the next true start (>= e: done) or, with *err set, the position of the failure.
template <class Or, class Match>
TK_HD uint32_t tk_rx_resolve_step_with(const TkRxProg& P, TkRxText& t, const TkRxMaps& M, uint32_t p, uint32_t e, Or&& orbits, uint32_t* err, Match&& match) {
    const TkRxPlan R = tk_rx_plan(M, t.n, p >> M.seg_shift, p, e);
    if (R.ok) {
        const uint32_t last = tk_rx_emit(M, R, p, orbits);
        if (R.exit != TK_RX_UNKNOWN) return R.exit;
        p = last;  // the guess stopped here (it would have had to look too far ahead): go on from its last start
    } else {
        orbits(p >> 5, 1u << (p & 31u), 0u);
    }
    bool gap;
    const uint32_t q = tk_rx_next_with(P, t, p, &gap, match);
    if (TK_RX_IS_ERROR(q)) {
        *err = q == TK_RX_OVERFLOW ? TK_RX_ERR_STACK : TK_RX_ERR_LIMIT;
        return p;
    }
    if (gap) orbits(p >> 5, 0u, 1u << (p & 31u));
    return q;
}
template <int DFA = 0, class Or>
TK_HD uint32_t tk_rx_resolve_step(const TkRxProg& P, TkRxText& t, const TkRxMaps& M, uint32_t p, uint32_t e, Or&& orbits, uint32_t* err) {
    return tk_rx_resolve_step_with(P, t, M, p, e, orbits, err, [&](uint32_t at) { return tk_rx_match_sel<DFA>(P, t, at); });
}

// One document [b, e) of the chunk by one lane.  `orbits(word index, start bits, gap bits)` ORs into the bitmaps of true starts and of the
// gap chars among them (shared words: atomic on the device).  Returns 0 or the error bits.
template <int DFA = 0, class Or>
TK_HD uint32_t tk_rx_resolve_lane(const TkRxProg& P, TkRxText t, const TkRxMaps& M, uint32_t b, uint32_t e, Or&& orbits, uint32_t* err_pos) {
    t.limit = 0xFFFFFFFFu;
    t.hit = false;
    uint32_t p = b, err = 0;
    while (p < e) {
        p = tk_rx_resolve_step<DFA>(P, t, M, p, e, orbits, &err);
        if (err) {
            *err_pos = p;
            return err;
        }
    }
    return 0;
}

// The same by a group of TK_RX_WAVE lanes that look at consecutive segments at once: lane j plans segment k0 + j as if the chain entered
// it at xexit[k0 + j - 1] (lane 0: at p); the longest prefix of lanes whose plans hold and whose exits are what the next lane assumed is
// taken in one go, and the chain continues behind it.  Where lane 0 has no plan the group takes one step of the serial form.  The host
// form runs the lanes one after the other (the CPU tests); the device form (tk_regex_kernels.h) ballots.
#define TK_RX_WAVE 64u
template <int DFA = 0, class Or>
TK_HD uint32_t tk_rx_resolve_group_host(const TkRxProg& P, TkRxText t, const TkRxMaps& M, uint32_t b, uint32_t e, Or&& orbits, uint32_t* err_pos) {
    t.limit = 0xFFFFFFFFu;
    t.hit = false;
    uint32_t p = b, err = 0;
    while (p < e) {
        const uint32_t k0 = p >> M.seg_shift;
        TkRxPlan plan[TK_RX_WAVE];
        uint32_t entry[TK_RX_WAVE], L = 0;
        for (uint32_t j = 0; j < TK_RX_WAVE; ++j) {
            entry[j] = j ? (M.spec && ((uint64_t)(k0 + j) << M.seg_shift) < t.n ? M.xexit[k0 + j - 1] : TK_RX_UNKNOWN) : p;
            plan[j] = entry[j] == TK_RX_UNKNOWN ? TkRxPlan{false, false, 0, 0, TK_RX_UNKNOWN} : tk_rx_plan(M, t.n, k0 + j, entry[j], e);
        }
        // lanes 0 .. L - 1: every plan holds, every exit but the last is known and is the next lane's entry
        while (L < TK_RX_WAVE && plan[L].ok && (L == 0 || (plan[L - 1].exit != TK_RX_UNKNOWN && plan[L - 1].exit == entry[L]))) ++L;
        if (L && plan[L - 1].exit == TK_RX_UNKNOWN) {  // (the last plan's chain broke off: its segment is left to the serial step)
            --L;
        }
        if (L == 0) {
            if constexpr (DFA) {  // the group matches the piece together (the host form runs the lanes of a scan one after the other)
                auto coop = [&](uint32_t S, uint32_t start, uint32_t pos, uint32_t base, uint32_t* pbad, uint32_t* m1, uint32_t* pnext) {
                    uint32_t bad[TK_RX_COOP_LANES], mat[TK_RX_COOP_LANES], pb = TK_RX_NONE, m = 0;
                    for (uint32_t j = 0; j < TK_RX_COOP_LANES; ++j) {
                        tk_rx_run_lane(P, t, S, start, pos, base + 16u * j, &bad[j], &mat[j], pnext);  // (*pnext: the last lane's)
                        pb = bad[j] < pb ? bad[j] : pb;
                    }
                    for (uint32_t j = 0; j < TK_RX_COOP_LANES; ++j)
                        if (mat[j] != TK_RX_NONE && mat[j] < pb && mat[j] + 1u > m) m = mat[j] + 1u;
                    *pbad = pb;
                    *m1 = m;
                };
                p = tk_rx_resolve_step_with(P, t, M, p, e, orbits, &err, [&](uint32_t at) { return tk_rx_match_dfa_coop<DFA == TK_RX_M_DFA_PREV>(P, t, at, coop); });
            } else {
                p = tk_rx_resolve_step<DFA>(P, t, M, p, e, orbits, &err);
            }
            if (err) {
                *err_pos = p;
                return err;
            }
            continue;
        }
        for (uint32_t j = 0; j < L; ++j) (void)tk_rx_emit(M, plan[j], entry[j], orbits);
        p = plan[L - 1].exit;
    }
    return 0;
}
