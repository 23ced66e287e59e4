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
// A speculative match never reads more than TK_RX_AHEAD bytes beyond its segment (a megabyte of one letter is one piece: every lane inside
// it would scan to its end); a lane that would have to stops, and the resolving lane -- which really is at a piece start -- goes on.
//
// Special tokens (encode() with allowed_special): a haystack ends where a special token starts (`hard` bit, as at a document start);
// the token itself is one step of the chain.  A position where the pattern does not match: the reference's find_iter goes on to the next
// match and the text in between yields no token (src/lib.rs:365,405).  Here such a char is a step of the chain as well -- a GAP piece of
// one char, marked in a second bitmap (`sgap` beside `spec`, `ggap` beside the true starts) -- and the pipeline behind the split gives a
// gap piece no token (TK_RES_GAP, tk_fused.h).
#pragma once
#include "tk_regex.h"

// bytes per speculative segment = 1 << seg_shift (a multiple of 32: a segment owns its words of the bitmap): 256 for chunks that would
// not fill the GPU with 1 KiB segments, 1024 otherwise (fewer, longer chains: less is matched twice around the segment boundaries)
#define TK_RX_SEG_SHIFT_SMALL 8u
#define TK_RX_SEG_SHIFT_LARGE 10u
#define TK_RX_SEG_SMALL_BELOW (256ull << 20)  // chunk bytes
#define TK_RX_AHEAD 16384u  // bytes a speculative match may look beyond its segment
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
    // the text word and the bitmap word read last: a lane walks its text byte by byte, one load per four bytes / thirty-two positions
    uint32_t tw_at = 0xFFFFFFFFu, tw = 0, bw_at = 0xFFFFFFFFu, bw = 0;
    TK_HD uint32_t byte(uint32_t p) {
        const uint32_t i = p >> 2;
        if (i != tw_at) {
            tw_at = i;
#if defined(__HIP_DEVICE_COMPILE__)
            tw = *(const uint32_t*)(text + 4u * (size_t)i);  // (the chunk's text is 16-byte aligned and readable 64 bytes past n)
#else
            __builtin_memcpy(&tw, text + 4u * (size_t)i, 4);
#endif
        }
        return (tw >> (8u * (p & 3u))) & 0xFFu;
    }
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

// the start that follows the piece (or special token, or gap char) that starts at p; or an error code (TK_RX_IS_ERROR: stack, budget)
TK_HD uint32_t tk_rx_next(const TkRxProg& P, TkRxText& t, uint32_t p, bool* gap) {
    *gap = false;
    if (t.special(p)) {
        uint32_t q = p + 1;
        while (q < t.n && !t.hard(q)) ++q;
        return q;
    }
    TK_RX_ON_MATCH();
    const uint32_t e = tk_rx_match(P, t, p);
    if (e != TK_RX_FAILED) return e;
    // no match starts here: the char is skipped (find_iter tries the next position)
    *gap = true;
    uint32_t q = p + 1;
    while (q < t.n && !t.hard(q) && (t.byte(q) & 0xC0u) == 0x80u) ++q;
    return q;
}

TK_HD void tk_rx_speculate_lane(const TkRxProg& P, TkRxText t, uint32_t k, uint32_t seg_shift, uint32_t* spec, uint32_t* sgap, uint32_t* xexit) {
    const uint64_t a64 = (uint64_t)k << seg_shift;
    if (a64 >= t.n) return;
    const uint32_t seg = 1u << seg_shift;
    const uint32_t a = (uint32_t)a64, end = t.n - a > seg ? a + seg : t.n;
    t.limit = t.n - end > TK_RX_AHEAD ? end + TK_RX_AHEAD : t.n;
    t.hit = false;
    uint32_t p = a;
    while (p < end && ((t.byte(p) & 0xC0u) == 0x80u || t.inside_special(p))) ++p;
    uint32_t x = TK_RX_UNKNOWN;
    if (p < end) {
        for (;;) {
            spec[p >> 5] |= 1u << (p & 31u);  // (the words of a segment belong to its lane)
            bool gap;
            const uint32_t q = tk_rx_next(P, t, p, &gap);
            if (TK_RX_IS_ERROR(q) || t.hit) break;
            if (gap) sgap[p >> 5] |= 1u << (p & 31u);
            if (q >= end) {
                x = q;
                break;
            }
            p = q;
        }
    }
    xexit[k] = x;
}

// One document [b, e) of the chunk.  `orbits(word index, start bits, gap bits)` ORs into the bitmaps of true starts and of the gap chars
// among them (shared words: atomic on the device).  Returns 0 or the error bits.
template <class Or>
TK_HD uint32_t tk_rx_resolve_lane(const TkRxProg& P, TkRxText t, uint32_t b, uint32_t e, uint32_t seg_shift, const uint32_t* spec, const uint32_t* sgap,
                                  const uint32_t* xexit, Or&& orbits, uint32_t* err_pos) {
    t.limit = 0xFFFFFFFFu;
    t.hit = false;
    uint32_t p = b;
    while (p < e) {
        bool run = true;
        if (spec && ((spec[p >> 5] >> (p & 31u)) & 1u)) {  // on segment k's chain: its bits from p on are true starts
            const uint32_t k = p >> seg_shift;
            const uint64_t se64 = ((uint64_t)k + 1u) << seg_shift;
            const uint32_t seg_end = se64 < e ? (uint32_t)se64 : e;
            uint32_t last = p;
            for (uint32_t w = p >> 5; w <= (seg_end - 1u) >> 5; ++w) {
                uint32_t bits = spec[w];
                if (w == (p >> 5)) bits &= ~0u << (p & 31u);
                if (w == ((seg_end - 1u) >> 5) && (seg_end & 31u)) bits &= (1u << (seg_end & 31u)) - 1u;
                if (bits) {
                    last = w * 32u + 31u - (uint32_t)__builtin_clz(bits);
                    orbits(w, bits, sgap[w] & bits);  // (a start whose evaluation the guess broke off has no gap bit: it is matched again below)
                }
            }
            const uint32_t x = xexit[k];
            if (x != TK_RX_UNKNOWN) {
                p = x;
                run = false;
            } else {
                p = last;  // the guess stopped here (it would have had to look too far ahead): go on from its last start
            }
        } else {
            orbits(p >> 5, 1u << (p & 31u), 0u);
        }
        if (run) {
            bool gap;
            const uint32_t q = tk_rx_next(P, t, p, &gap);
            if (TK_RX_IS_ERROR(q)) {
                *err_pos = p;
                return q == TK_RX_OVERFLOW ? TK_RX_ERR_STACK : TK_RX_ERR_LIMIT;
            }
            if (gap) orbits(p >> 5, 0u, 1u << (p & 31u));
            p = q;
        }
    }
    return 0;
}
