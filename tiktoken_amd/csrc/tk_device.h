// Device functions of the BPE encode path (host-compilable for the CPU-side unit tests of the
// logic; the kernels in tk_kernels.hip are the only product callers).
//
//   tk_classify_*      UTF-8 decode + two-stage Unicode class table (what \p{L}, \p{N}, \p{M}, \s
//                      mean to the regex at reference src/lib.rs:365)
//   tk_piece_end       the pre-tokeniser: end of the regex match that starts at p, for the three
//                      stock patterns (tiktoken_ext/openai_public.py:12-14, :89, :104-114)
//   tk_certain_start   class pairs after which a piece boundary is certain whatever precedes
//   tk_probe_piece / tk_probe_pair   exact table probes (src/lib.rs:367, :150, :165-167)
#pragma once
#include "tk_common.h"

// ------------------------------------------------------------------------------------------
// classification
// ------------------------------------------------------------------------------------------
TK_HD uint32_t tk_class_of_cp(const TkTables& T, uint32_t cp) {
    if (cp > 0x10FFFFu) return TK_C_OT;
    return T.uc_stage2[(uint32_t)T.uc_stage1[cp >> 8] * 256u + (cp & 255u)];
}

// Class of the text byte at `pos` (TK_C_CONT for continuation bytes).  `n` bounds the reads.
TK_HD uint32_t tk_classify_text(const TkTables& T, const uint8_t* __restrict__ text, uint64_t pos, uint64_t n) {
    uint32_t b = text[pos];
    if (b < 0x80u) return tk_class_of_cp(T, b);
    if ((b & 0xC0u) == 0x80u) return TK_C_CONT;
    uint32_t len = b >= 0xF0u ? 4u : (b >= 0xE0u ? 3u : 2u);
    if (pos + len > n) return TK_C_OT;  // truncated sequence (invalid UTF-8): treated as OTHER
    uint32_t cp;
    if (len == 2u)
        cp = ((b & 0x1Fu) << 6) | (text[pos + 1] & 0x3Fu);
    else if (len == 3u)
        cp = ((b & 0x0Fu) << 12) | ((uint32_t)(text[pos + 1] & 0x3Fu) << 6) | (text[pos + 2] & 0x3Fu);
    else
        cp = ((b & 0x07u) << 18) | ((uint32_t)(text[pos + 1] & 0x3Fu) << 12) |
             ((uint32_t)(text[pos + 2] & 0x3Fu) << 6) | (text[pos + 3] & 0x3Fu);
    return tk_class_of_cp(T, cp);
}

TK_HD bool tk_bit(const uint32_t* __restrict__ bm, uint64_t pos) { return (bm[pos >> 5] >> (pos & 31u)) & 1u; }

// Class byte (class | flags) of position pos < n, combining the text class with the break and
// special-token bitmaps (either of the latter may be null).
TK_HD uint32_t tk_class_byte(const TkTables& T, const uint8_t* __restrict__ text, uint64_t pos, uint64_t n,
                             const uint32_t* __restrict__ brk, const uint32_t* __restrict__ spec_start,
                             const uint32_t* __restrict__ spec_in) {
    if (spec_in && tk_bit(spec_in, pos)) return TK_C_CONT;
    if (spec_start && tk_bit(spec_start, pos)) return TK_C_SPEC | TK_F_HARD;
    uint32_t c = tk_classify_text(T, text, pos, n);
    if (c != TK_C_CONT && brk && tk_bit(brk, pos)) c |= TK_F_HARD;
    return c;
}

// ------------------------------------------------------------------------------------------
// Certain piece starts.  CERT[pat][a] is the set of classes b such that a char of class b that
// follows a char of class a ALWAYS starts a new piece, whatever the left context.  Derived by
// exhaustive comparison against Python `regex` (tests/test_oracle.py re-derives it), e.g. a
// letter is never followed inside its piece by white space, and digits only ever share a piece
// with digits.  They make work units independent: any scanner started at a certain start is in
// phase with the sequential regex.
// ------------------------------------------------------------------------------------------
#define TK_ALLC (TK_CB(TK_C_NL) | TK_CB(TK_C_SP) | TK_CB(TK_C_WSO) | TK_M_L | TK_M_OTHER | TK_CB(TK_C_NU))
// mask of classes b that certainly start a piece after class a
TK_HD uint32_t tk_certain_mask(int pat, uint32_t a) {
    uint32_t m = 0;
    const uint32_t WS3 = TK_M_WS, NU = TK_CB(TK_C_NU);
    if (pat == TK_PAT_R50K) {
        switch (a) {
            case TK_C_NL: case TK_C_WSO: m = TK_M_L | TK_M_OTHER | NU; break;
            case TK_C_LU: case TK_C_LL: case TK_C_LC: m = WS3 | TK_M_OTHER | NU; break;
            case TK_C_MK: case TK_C_SL: case TK_C_OT: m = WS3 | TK_M_L | NU; break;
            case TK_C_AP: m = WS3 | TK_CB(TK_C_LU) | TK_CB(TK_C_LC) | NU; break;
            case TK_C_NU: m = TK_ALLC & ~NU; break;
            default: m = 0;
        }
    } else if (pat == TK_PAT_CL100K) {
        switch (a) {
            case TK_C_NL: m = TK_M_L | TK_M_OTHER | NU; break;
            case TK_C_SP: m = NU; break;
            case TK_C_WSO: m = TK_M_OTHER | NU; break;
            case TK_C_LU: case TK_C_LL: case TK_C_LC: m = WS3 | TK_M_OTHER | NU; break;
            case TK_C_MK: case TK_C_AP: case TK_C_SL: case TK_C_OT: m = TK_CB(TK_C_SP) | TK_CB(TK_C_WSO) | NU; break;
            case TK_C_NU: m = TK_ALLC & ~NU; break;
            default: m = 0;
        }
    } else {
        switch (a) {
            case TK_C_NL: m = TK_M_L | TK_CB(TK_C_MK) | NU | TK_CB(TK_C_AP) | TK_CB(TK_C_OT); break;
            case TK_C_SP: m = NU; break;
            case TK_C_WSO: m = NU | TK_CB(TK_C_AP) | TK_CB(TK_C_SL) | TK_CB(TK_C_OT); break;
            // (a lower-case letter followed by an upper-case one is NOT certain: "'lL" is a contraction)
            case TK_C_LU: case TK_C_LL: case TK_C_LC: m = WS3 | NU | TK_CB(TK_C_SL) | TK_CB(TK_C_OT); break;
            case TK_C_MK: case TK_C_AP: case TK_C_SL: case TK_C_OT: m = TK_CB(TK_C_SP) | TK_CB(TK_C_WSO) | NU; break;
            case TK_C_NU: m = TK_ALLC & ~NU; break;
            default: m = 0;
        }
    }
    return m;
}
TK_HD bool tk_certain_start(int pat, uint32_t a, uint32_t b) { return (tk_certain_mask(pat, a) >> b) & 1u; }

// The opposite table: NEVER[pat][a] = classes b such that NO piece of the stock pattern starts at a char of class b that follows a char
// of class a -- unless an apostrophe stands two or three bytes before it (the end of a contraction is a boundary inside a run of
// letters).  Derived by brute force against Python `regex` like the table above; the front kernel uses it to take the common case
// "the piece ends at the next certain start" without running the scanner (tk_chunk_never, tk_chunk.h), and the CPU simulation checks
// every piece start of the test corpora against it.
TK_HD uint32_t tk_never_mask(int pat, uint32_t a) {
    const uint32_t L3 = TK_M_L, MK = TK_CB(TK_C_MK), NL = TK_CB(TK_C_NL), O4 = TK_M_OTHER;
    if (pat == TK_PAT_R50K) {
        switch (a) {
            case TK_C_SP: return L3 | O4 | TK_CB(TK_C_NU);
            case TK_C_LU: case TK_C_LL: case TK_C_LC: return L3;
            case TK_C_NU: return TK_CB(TK_C_NU);
            case TK_C_MK: case TK_C_AP: case TK_C_SL: case TK_C_OT: return O4;
            default: return 0;
        }
    } else if (pat == TK_PAT_CL100K) {
        switch (a) {
            case TK_C_NL: return NL;
            case TK_C_SP: return NL | L3 | O4;
            case TK_C_WSO: return NL | L3;
            case TK_C_LU: case TK_C_LL: case TK_C_LC: return L3;
            case TK_C_MK: case TK_C_AP: case TK_C_SL: case TK_C_OT: return NL | O4;
            default: return 0;
        }
    } else {
        switch (a) {
            case TK_C_NL: return NL;
            case TK_C_SP: return NL | L3 | O4;
            case TK_C_WSO: return NL | L3 | MK;
            case TK_C_LU: return L3 | MK;
            case TK_C_LL: case TK_C_LC: return TK_CB(TK_C_LL) | TK_CB(TK_C_LC) | MK;  // (an upper-case letter may start the next word)
            case TK_C_MK: return MK;
            case TK_C_AP: case TK_C_OT: return NL | O4;
            case TK_C_SL: return NL | TK_CB(TK_C_SL);
            default: return 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The scanner.  `A` is an accessor: A::cls(pos) -> class byte of position pos (TK_C_END at and
// beyond the end of the buffer), A::byte(pos) -> raw text byte.
// ------------------------------------------------------------------------------------------
template <class A>
TK_HD uint32_t tk_la(A& a, uint64_t pos) {  // look-ahead class: a hard start looks like end-of-text
    uint32_t c = a.cls(pos);
    return (c & TK_F_HARD) ? (uint32_t)TK_C_END : (c & 15u);
}

template <class A>
TK_HD uint64_t tk_next_char(A& a, uint64_t pos) {  // pos is a char start; returns the next char start
    ++pos;
    while (a.cls(pos) == TK_C_CONT) ++pos;
    return pos;
}

template <class A>
TK_HD uint64_t tk_run_end(A& a, uint64_t s, uint32_t mask) {
    while ((mask >> tk_la(a, s)) & 1u) s = tk_next_char(a, s);
    return s;
}

// byte length of a contraction whose apostrophe is at p, or 0.  The list comes with the pattern (TkPat); case-insensitive forms fold
// U+017F to 's', as Unicode simple case folding does in both regex engines (no other char folds to a letter the parser admits).
template <class A>
TK_HD uint32_t tk_contraction_len(A& a, uint64_t p, TkPat pat) {
    if (tk_la(a, p + 1) == TK_C_END) return 0;
    const uint32_t b1 = a.byte(p + 1);
    const bool ci = pat.ci();
    if (ci && b1 == 0xC5u) return (((pat.c1 >> ('s' - 'a')) & 1u) && a.cls(p + 2) == TK_C_CONT && a.byte(p + 2) == 0xBFu) ? 3u : 0u;
    const uint32_t al = ci ? (b1 | 0x20u) : b1;
    if (b1 >= 0x80u || al < 'a' || al > 'z') return 0;
    if ((pat.c1 >> (al - 'a')) & 1u) return 2;
    if (!pat.n2() || tk_la(a, p + 2) == TK_C_END) return 0;
    const uint32_t b2 = a.byte(p + 2), bl = ci ? (b2 | 0x20u) : b2;
    if (b2 >= 0x80u) return 0;
    const uint32_t key = (al << 8) | bl;
    for (uint32_t i = 0; i < pat.n2(); ++i)
        if (pat.two(i) == key) return 3;
    return 0;
}

// \s++$ | \s*[\r\n]+? | \s+(?!\S) | \s  -- p is a white-space char
template <class A>
TK_HD uint64_t tk_ws_tail(A& a, uint64_t p, TkPat pat) {
    uint64_t q = p, last_start = p, after_last_nl = 0;
    uint32_t nchars = 0;
    bool has_nl = false;
    uint32_t c = a.cls(p) & 15u;  // p itself is a piece start: its own hard-start flag is not an end
    for (;;) {
        if (!((TK_M_WS >> c) & 1u)) break;
        last_start = q;
        q = tk_next_char(a, q);
        ++nchars;
        if (c == TK_C_NL) {
            has_nl = true;
            after_last_nl = q;
        }
        c = tk_la(a, q);
    }
    bool at_end = (c == TK_C_END);
    if (pat.ws_dollar() && at_end) return q;
    if (pat.nl_rule() && has_nl) return after_last_nl;
    if (at_end) return q;
    if (nchars >= 2) return last_start;
    return q;
}

// o200k letter alternatives from s (s is the first letter-ish char; its flags were already
// handled by the caller)
template <class A>
TK_HD uint64_t tk_o200k_word(A& a, uint64_t s, uint32_t c_first, TkPat pat) {
    uint64_t r_end = s, after_last_c = 0;
    bool has_c = false;
    uint32_t c = c_first;
    while ((TK_M_UPPERISH >> c) & 1u) {
        r_end = tk_next_char(a, r_end);
        if (c != TK_C_LU) {
            has_c = true;
            after_last_c = r_end;
        }
        c = tk_la(a, r_end);
    }
    uint64_t t_end = r_end;
    while ((TK_M_LOWERISH >> c) & 1u) {
        t_end = tk_next_char(a, t_end);
        c = tk_la(a, t_end);
    }
    uint64_t e;
    if (t_end > r_end) {
        e = t_end;
    } else if (has_c) {
        e = after_last_c;
        c = tk_la(a, e);
    } else {
        e = r_end;
    }
    if (c == TK_C_AP) e += tk_contraction_len(a, e, pat);
    return e;
}

// \p{N}{1,k} from p (k = 0: \p{N}+)
template <class A>
TK_HD uint64_t tk_digits(A& a, uint64_t p, uint32_t k) {
    uint64_t e = tk_next_char(a, p);
    for (uint32_t i = 1; k == 0u || i < k; ++i) {
        if (tk_la(a, e) != TK_C_NU) break;
        e = tk_next_char(a, e);
    }
    return e;
}

// End (exclusive) of the piece that starts at p.  p must be a true piece start.
template <class A>
TK_HD uint64_t tk_piece_end(A& a, uint64_t p, TkPat pat) {
    uint32_t c = a.cls(p) & 15u;
    uint64_t p1 = tk_next_char(a, p);
    if (c == TK_C_SPEC) return p1;  // a whole special token (its interior bytes are TK_C_CONT)
    uint32_t nxt = tk_la(a, p1);
    if (pat.fam() == TK_PAT_R50K) {
        if (c == TK_C_AP) {
            uint32_t k = tk_contraction_len(a, p, pat);
            if (k) return p + k;
        }
        uint64_t s = p;
        uint32_t k = c;
        if (c == TK_C_SP && nxt != TK_C_END) {
            s = p1;
            k = nxt;
        }
        if ((TK_M_L >> k) & 1u) return tk_run_end(a, tk_next_char(a, s), TK_M_L);
        if (k == TK_C_NU) return tk_run_end(a, tk_next_char(a, s), TK_CB(TK_C_NU));
        if ((TK_M_OTHER >> k) & 1u) return tk_run_end(a, tk_next_char(a, s), TK_M_OTHER);
        return tk_ws_tail(a, p, pat);
    }
    if (pat.fam() == TK_PAT_CL100K) {
        if (c == TK_C_AP) {
            uint32_t k = tk_contraction_len(a, p, pat);
            if (k) return p + k;
        }
        if (((TK_M_L >> c) & 1u) || (c != TK_C_NL && c != TK_C_NU && ((TK_M_L >> nxt) & 1u)))
            return tk_run_end(a, p1, TK_M_L);
        if (c == TK_C_NU) return tk_digits(a, p, pat.digits());
        uint64_t s = p;
        uint32_t k = c;
        if (c == TK_C_SP && nxt != TK_C_END) {
            s = p1;
            k = nxt;
        }
        if ((TK_M_OTHER >> k) & 1u) {
            uint64_t e = tk_run_end(a, tk_next_char(a, s), TK_M_OTHER);
            return tk_run_end(a, e, pat.suffix_mask());
        }
        return tk_ws_tail(a, p, pat);
    }
    // o200k
    if ((TK_M_WORD >> c) & 1u) return tk_o200k_word(a, p, c, pat);
    if (c != TK_C_NL && c != TK_C_NU && ((TK_M_WORD >> nxt) & 1u)) return tk_o200k_word(a, p1, nxt, pat);
    if (c == TK_C_NU) return tk_digits(a, p, pat.digits());
    uint64_t s = p;
    uint32_t k = c;
    if (c == TK_C_SP && nxt != TK_C_END) {
        s = p1;
        k = nxt;
    }
    if ((TK_M_OTHER >> k) & 1u) {
        uint64_t e = tk_run_end(a, tk_next_char(a, s), TK_M_OTHER);
        return tk_run_end(a, e, pat.suffix_mask());
    }
    return tk_ws_tail(a, p, pat);
}

// ------------------------------------------------------------------------------------------
// The same scanner written over RUN QUERIES instead of a byte walk: for pieces that leave a tile's LDS window (a run of a million
// letters, spaces or CJK chars is one piece: reference tests/test_encoding.py:52-57,113-124) the front kernel answers each query
// with the whole workgroup, 4 KiB per step, so that such a piece costs microseconds per KiB instead of microseconds per byte.
// R provides  cls(pos), byte(pos)                  as the accessors above (single positions)
//             run_end(from, mask)                  = tk_run_end: first char start s >= from whose look-ahead class is not in mask
//             last_in(from, to, mask)              start of the last char of [from, to) whose class nibble is in mask, or ~0
// Must return exactly what tk_piece_end returns (checked on the CPU for every piece of the test corpora).
// ------------------------------------------------------------------------------------------
#define TK_NO_POS 0xFFFFFFFFFFFFFFFFull
template <class R>
TK_HD uint64_t tk_ws_tail_runs(R& r, uint64_t p, TkPat pat) {
    const uint32_t c0 = r.cls(p) & 15u;
    if (!((TK_M_WS >> c0) & 1u)) return p;
    const uint64_t p1 = tk_next_char(r, p);
    const uint64_t q = r.run_end(p1, TK_M_WS);
    const bool at_end = tk_la(r, q) == TK_C_END;
    if (pat.ws_dollar() && at_end) return q;
    if (pat.nl_rule()) {
        const uint64_t nlp = r.last_in(p, q, TK_CB(TK_C_NL));
        if (nlp != TK_NO_POS) return nlp + 1;  // (\r and \n are single bytes)
    }
    if (at_end) return q;
    if (q > p1) return r.last_in(p, q, TK_M_WS);  // two or more chars: back off one
    return q;
}
template <class R>
TK_HD uint64_t tk_o200k_word_runs(R& r, uint64_t s, uint32_t c_first, TkPat pat) {
    uint64_t r_end = s;
    if ((TK_M_UPPERISH >> c_first) & 1u) r_end = r.run_end(tk_next_char(r, s), TK_M_UPPERISH);
    uint32_t c = r_end == s ? c_first : tk_la(r, r_end);
    uint64_t t_end = r_end;
    if ((TK_M_LOWERISH >> c) & 1u) {
        t_end = r.run_end(tk_next_char(r, r_end), TK_M_LOWERISH);
        c = tk_la(r, t_end);
    }
    uint64_t e;
    if (t_end > r_end) {
        e = t_end;
    } else {
        const uint64_t lc = r_end > s ? r.last_in(s, r_end, TK_CB(TK_C_LC) | TK_CB(TK_C_MK)) : TK_NO_POS;
        if (lc != TK_NO_POS) {
            e = tk_next_char(r, lc);
            c = tk_la(r, e);
        } else {
            e = r_end;
        }
    }
    if (c == TK_C_AP) e += tk_contraction_len(r, e, pat);
    return e;
}
template <class R>
TK_HD uint64_t tk_piece_end_runs(R& r, uint64_t p, TkPat pat) {
    const uint32_t c = r.cls(p) & 15u;
    const uint64_t p1 = tk_next_char(r, p);
    if (c == TK_C_SPEC) return p1;
    const uint32_t nxt = tk_la(r, p1);
    if (pat.fam() == TK_PAT_O200K) {
        if ((TK_M_WORD >> c) & 1u) return tk_o200k_word_runs(r, p, c, pat);
        if (c != TK_C_NL && c != TK_C_NU && ((TK_M_WORD >> nxt) & 1u)) return tk_o200k_word_runs(r, p1, nxt, pat);
        if (c == TK_C_NU) return tk_digits(r, p, pat.digits());
    } else {
        if (c == TK_C_AP) {
            const uint32_t k = tk_contraction_len(r, p, pat);
            if (k) return p + k;
        }
        if (pat.fam() == TK_PAT_CL100K) {
            if (((TK_M_L >> c) & 1u) || (c != TK_C_NL && c != TK_C_NU && ((TK_M_L >> nxt) & 1u))) return r.run_end(p1, TK_M_L);
            if (c == TK_C_NU) return tk_digits(r, p, pat.digits());
        }
    }
    uint64_t s = p;
    uint32_t k = c;
    if (c == TK_C_SP && nxt != TK_C_END) {
        s = p1;
        k = nxt;
    }
    if (pat.fam() == TK_PAT_R50K) {
        if ((TK_M_L >> k) & 1u) return r.run_end(tk_next_char(r, s), TK_M_L);
        if (k == TK_C_NU) return r.run_end(tk_next_char(r, s), TK_CB(TK_C_NU));
        if ((TK_M_OTHER >> k) & 1u) return r.run_end(tk_next_char(r, s), TK_M_OTHER);
    } else if ((TK_M_OTHER >> k) & 1u) {
        const uint64_t e = r.run_end(tk_next_char(r, s), TK_M_OTHER);
        return r.run_end(e, pat.suffix_mask());
    }
    return tk_ws_tail_runs(r, p, pat);
}

// ------------------------------------------------------------------------------------------
// Bit-parallel form of the scanner.  The pre-tokeniser kernel keeps, per tile, one bitmap per class
// set (bit = text byte; continuation bytes carry the class of their char).  At a piece start p it
// extracts 64-bit windows (bit k <-> position p + k); run ends are then `ctz` of a masked window
// instead of a byte-walking loop.  Returns the piece length in bytes, or 0 when the piece is not
// resolved inside the window (caller falls back to tk_piece_end).
// ------------------------------------------------------------------------------------------
enum { TKB_START = 0, TKB_HARD, TKB_L, TKB_UP, TKB_LOW, TKB_CAS, TKB_OTH, TKB_WS, TKB_NL, TKB_NU, TKB_NLSL, TKB_KINDS };
// A window provider has `start`, `stop` and get(kind) for the class-set bitmaps; the scanner only asks for the
// kinds its branch needs (the kernel's provider extracts them from the LDS bitmaps on demand).
struct TkWin {
    uint64_t start;  // char starts
    uint64_t stop;   // positions k >= 1 where look-ahead sees end-of-text (hard start or past the end)
    uint64_t L, up, low, cas, oth, ws, nl, nu, nlsl;
    TK_HD uint64_t get(int kind) const {
        switch (kind) {
            case TKB_L: return L;
            case TKB_UP: return up;
            case TKB_LOW: return low;
            case TKB_CAS: return cas;
            case TKB_OTH: return oth;
            case TKB_WS: return ws;
            case TKB_NL: return nl;
            case TKB_NU: return nu;
            default: return nlsl;
        }
    }
};

#if defined(__HIP_DEVICE_COMPILE__)
TK_HD uint32_t tk_ctz64(uint64_t v) { return v ? (uint32_t)(__ffsll((unsigned long long)v) - 1) : 64u; }
TK_HD uint32_t tk_clz64(uint64_t v) { return v ? (uint32_t)__clzll((long long)v) : 64u; }
TK_HD uint32_t tk_popc64(uint64_t v) { return (uint32_t)__popcll(v); }
#else
TK_HD uint32_t tk_ctz64(uint64_t v) { return v ? (uint32_t)__builtin_ctzll(v) : 64u; }
TK_HD uint32_t tk_clz64(uint64_t v) { return v ? (uint32_t)__builtin_clzll(v) : 64u; }
TK_HD uint32_t tk_popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
#endif

TK_HD uint64_t tk_below(uint32_t b) { return b >= 64u ? ~0ull : ((1ull << b) - 1ull); }  // bits [0, b)
// number of consecutive positions k >= from whose bit is set and that are not stops
TK_HD uint32_t tk_run(uint64_t bits, uint64_t stop, uint32_t from) {
    if (from >= 64u) return 0;
    uint64_t x = (bits & ~stop) >> from;
    return tk_ctz64(~x);  // zeros shifted in from the top bound the result by 64 - from
}

#define TK_WIN_SAFE 58u  // results up to here are decided inside the first 64-bit window
#define TK_UNRES 0xFFFFFFFFu
// Runs that leave the first window continue through an extension provider X:
//   x.win(kind, j) -> 64-bit window j (positions [64 j, 64 j + 64) relative to the piece start) of bitmap `kind`
//   x.limit()      -> number of positions, counted from the piece start, for which windows are valid
// A run that reaches the limit is unresolved (TK_UNRES) and the caller falls back to tk_piece_end.
template <class X>
TK_HD uint32_t tk_runx(uint64_t bits0, uint64_t stop0, X& x, int kind, uint32_t from) {
    uint32_t pos = from;
    if (from < 64u) {
        uint32_t r = tk_run(bits0, stop0, from);
        if (from + r < 64u) return r;
        pos = 64u;
    }
    for (;;) {
        if (pos + 64u > x.limit()) return TK_UNRES;
        const uint32_t j = pos >> 6, o = pos & 63u;
        const uint64_t bits = (x.win(kind, j) & ~x.win(TKB_HARD, j)) >> o;
        const uint32_t r = tk_ctz64(~bits);
        pos += r;
        if (r < 64u - o) break;
    }
    return pos - from;
}
// highest set position of bitmap `kind` in [a, b), or -1 (b may lie beyond the first window)
template <class X>
TK_HD int tk_last_setx(uint64_t bits0, X& x, int kind, uint32_t a, uint32_t b) {
    if (b <= a) return -1;
    int j = (int)((b - 1u) >> 6);
    const int ja = (int)(a >> 6);
    for (; j >= ja; --j) {
        uint64_t w = j == 0 ? bits0 : x.win(kind, (uint32_t)j);
        const uint32_t hi = (uint32_t)j * 64u + 64u, lo = (uint32_t)j * 64u;
        if (b < hi) w &= tk_below(b - lo);
        if (a > lo) w &= ~tk_below(a - lo);
        if (w) return (int)(lo + 63u - tk_clz64(w));
    }
    return -1;
}
template <class X>
TK_HD bool tk_bitx(uint64_t bits0, X& x, int kind, uint32_t k) {
    return k < 64u ? (bits0 >> k) & 1ull : (x.win(kind, k >> 6) >> (k & 63u)) & 1ull;
}

template <class W, class A>
TK_HD uint32_t tk_contraction_bits(const W& w, A& a, uint64_t p, uint32_t e, TkPat pat) {
    if ((w.stop >> (e + 1)) & 1ull) return 0;
    const uint32_t b1 = a.byte(p + e + 1);
    const bool ci = pat.ci();
    if (ci && b1 == 0xC5u)
        return (((pat.c1 >> ('s' - 'a')) & 1u) && !((w.start >> (e + 2)) & 1ull) && !((w.stop >> (e + 2)) & 1ull) && a.byte(p + e + 2) == 0xBFu) ? 3u : 0u;
    const uint32_t al = ci ? (b1 | 0x20u) : b1;
    if (b1 >= 0x80u || al < 'a' || al > 'z') return 0;
    if ((pat.c1 >> (al - 'a')) & 1u) return 2;
    if (!pat.n2() || ((w.stop >> (e + 2)) & 1ull)) return 0;
    const uint32_t b2 = a.byte(p + e + 2), bl = ci ? (b2 | 0x20u) : b2;
    if (b2 >= 0x80u) return 0;
    const uint32_t key = (al << 8) | bl;
    for (uint32_t i = 0; i < pat.n2(); ++i)
        if (pat.two(i) == key) return 3;
    return 0;
}

// c = class nibble of the char at p.  Returns the piece length, or 0 if unresolved.
template <class W, class A, class X>
TK_HD uint32_t tk_piece_len_bits(const W& w, A& a, X& x, uint64_t p, uint32_t c, TkPat pat) {
    const uint64_t stop = w.stop;
    // length of the first char: next char start or stop after position 0
    const uint32_t k1 = 1u + tk_ctz64((w.start | stop) >> 1);
    if (k1 > 4u) return 0;  // (special-token pieces and anything odd go through the generic scanner)
    if (c == TK_C_SPEC) return 0;
    const bool nxt_end = (stop >> k1) & 1ull;
    uint32_t e = 0;
    if (pat.fam() == TK_PAT_O200K) {
        const uint64_t word = w.get(TKB_UP) | w.get(TKB_LOW);
        uint32_t ks = 64;
        if ((TK_M_WORD >> c) & 1u) ks = 0;
        else if (c != TK_C_NL && c != TK_C_NU && !nxt_end && ((word >> k1) & 1ull)) ks = k1;
        if (ks != 64u) {
            uint32_t r = tk_runx(w.get(TKB_UP), stop, x, TKB_UP, ks);
            if (r == TK_UNRES) return 0;
            const uint32_t re = ks + r;
            uint32_t t = tk_runx(w.get(TKB_LOW), stop, x, TKB_LOW, re);
            if (t == TK_UNRES) return 0;
            const uint32_t te = re + t;
            if (te > re) {
                e = te;
            } else if (re <= TK_WIN_SAFE) {
                uint64_t xx = w.get(TKB_CAS) & ~stop & tk_below(re) & ~tk_below(ks);
                e = xx ? 64u - tk_clz64(xx) : re;
            } else {
                int lc = tk_last_setx(w.get(TKB_CAS), x, TKB_CAS, ks, re);  // (stops cannot lie inside a run)
                e = lc >= 0 ? (uint32_t)lc + 1u : re;
            }
            if (e <= TK_WIN_SAFE) {
                if (!((stop >> e) & 1ull) && a.byte(p + e) == '\'') e += tk_contraction_bits(w, a, p, e, pat);
            } else if (e + 4u <= x.limit()) {
                if (!tk_bitx(w.stop, x, TKB_HARD, e) && a.byte(p + e) == '\'') e += tk_contraction_len(a, p + e, pat);
            } else {
                return 0;
            }
            return e;
        }
    } else if (pat.fam() == TK_PAT_CL100K) {
        if (c == TK_C_AP) {
            uint32_t k = tk_contraction_bits(w, a, p, 0, pat);
            if (k) return k;
        }
        if ((TK_M_L >> c) & 1u) {
            uint32_t r = tk_runx(w.get(TKB_L), stop, x, TKB_L, 0);
            return r == TK_UNRES ? 0 : r;
        }
        if (c != TK_C_NL && c != TK_C_NU && !nxt_end && ((w.get(TKB_L) >> k1) & 1ull)) {
            uint32_t r = tk_runx(w.get(TKB_L), stop, x, TKB_L, k1);
            return r == TK_UNRES ? 0 : k1 + r;
        }
    } else {
        if (c == TK_C_AP) {
            uint32_t k = tk_contraction_bits(w, a, p, 0, pat);
            if (k) return k;
        }
    }
    if (pat.fam() != TK_PAT_R50K && c == TK_C_NU) {
        uint32_t r = tk_run(w.get(TKB_NU), stop, 0);
        uint64_t sx = pat.digits() ? (w.start & tk_below(r)) : 0ull;  // start bits of the run's chars: the (k+1)-th one ends the group
        for (uint32_t i = 0; i < pat.digits(); ++i) {
            if (pat.generic() && !sx) break;
            sx &= sx - 1;
        }
        e = sx ? tk_ctz64(sx) : r;
        return e > TK_WIN_SAFE ? 0 : e;
    }
    // optional single space, then a run of one kind (r50k: letters / digits / other; others: other only)
    uint32_t s = 0;
    bool s_ok = true;
    if (c == TK_C_SP && !nxt_end) s = k1;
    if (pat.fam() == TK_PAT_R50K) {
        int kind = -1;
        uint64_t b0 = 0;
        if ((w.get(TKB_L) >> s) & 1ull) { kind = TKB_L; b0 = w.get(TKB_L); }
        else if ((w.get(TKB_NU) >> s) & 1ull) { kind = TKB_NU; b0 = w.get(TKB_NU); }
        else if ((w.get(TKB_OTH) >> s) & 1ull) { kind = TKB_OTH; b0 = w.get(TKB_OTH); }
        if (kind >= 0) {
            uint32_t r = tk_runx(b0, stop, x, kind, s);
            return r == TK_UNRES ? 0 : s + r;
        }
        s_ok = false;
    } else if ((w.get(TKB_OTH) >> s) & 1ull) {
        uint32_t r = tk_runx(w.get(TKB_OTH), stop, x, TKB_OTH, s);
        if (r == TK_UNRES) return 0;
        const uint32_t e1 = s + r;
        // (the suffix set behind the run: its own bitmap for o200k and for generic patterns, the newline bitmap for cl100k)
        uint32_t r2 = (pat.fam() == TK_PAT_O200K || pat.generic()) ? tk_runx(w.get(TKB_NLSL), stop, x, TKB_NLSL, e1) : tk_runx(w.get(TKB_NL), stop, x, TKB_NL, e1);
        return r2 == TK_UNRES ? 0 : e1 + r2;
    } else {
        s_ok = false;
    }
    (void)s_ok;
    // white space
    {
        uint32_t q = tk_runx(w.get(TKB_WS), stop, x, TKB_WS, 0);
        if (q == TK_UNRES) return 0;
        if (q <= TK_WIN_SAFE) {
            uint64_t rng = tk_below(q);
            bool at_end = (stop >> q) & 1ull;
            uint64_t nlr = w.get(TKB_NL) & rng, st = w.start & rng;
            if (pat.ws_dollar() && at_end) return q;
            if (pat.nl_rule() && nlr) return 64u - tk_clz64(nlr);
            if (at_end) return q;
            if (tk_popc64(st) >= 2u) return 63u - tk_clz64(st);
            return q;
        }
        if (q + 1u > x.limit()) return 0;
        const bool at_end = tk_bitx(w.stop, x, TKB_HARD, q);
        if (pat.ws_dollar() && at_end) return q;
        if (pat.nl_rule()) {
            int ln = tk_last_setx(w.get(TKB_NL), x, TKB_NL, 0, q);
            if (ln >= 0) return (uint32_t)ln + 1u;
        }
        if (at_end) return q;
        return (uint32_t)tk_last_setx(w.start, x, TKB_START, 0, q);  // a run this long has >= 2 chars
    }
}

// ------------------------------------------------------------------------------------------
// 32-bit fast path of the bit-parallel scanner.  Same rules as tk_piece_len_bits on windows of 32 positions;
// it answers only when every run involved ends by position TK_WIN32_SAFE (so that all look-ahead bits lie inside
// the window) and returns 0 otherwise -- the caller then runs the 64-bit scanner.  Most pieces are a few bytes
// long, and 32-bit funnels, shifts and ctz cost a third of their 64-bit forms on the vector ALU.
// W32 has uint32_t `start`, `stop` (bit 0 cleared) and get(kind).
// ------------------------------------------------------------------------------------------
#define TK_WIN32_SAFE 26u
#if defined(__HIP_DEVICE_COMPILE__)
TK_HD uint32_t tk_w32_ctz(uint32_t v) { return v ? (uint32_t)(__ffs((int)v) - 1) : 32u; }
TK_HD uint32_t tk_w32_clz(uint32_t v) { return v ? (uint32_t)__clz((int)v) : 32u; }
TK_HD uint32_t tk_w32_popc(uint32_t v) { return (uint32_t)__popc(v); }
#else
TK_HD uint32_t tk_w32_ctz(uint32_t v) { return v ? (uint32_t)__builtin_ctz(v) : 32u; }
TK_HD uint32_t tk_w32_clz(uint32_t v) { return v ? (uint32_t)__builtin_clz(v) : 32u; }
TK_HD uint32_t tk_w32_popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
#endif
TK_HD uint32_t tk_below32(uint32_t b) { return b >= 32u ? ~0u : ((1u << b) - 1u); }  // bits [0, b)
TK_HD uint32_t tk_run32(uint32_t bits, uint32_t stop, uint32_t from) {              // from <= TK_WIN32_SAFE
    return tk_w32_ctz(~((bits & ~stop) >> from));
}

template <class W32, class A>
TK_HD uint32_t tk_piece_len_bits32(const W32& w, A& a, uint64_t p, uint32_t c, TkPat pat) {
    const uint32_t stop = w.stop;
    const uint32_t k1 = 1u + tk_w32_ctz((w.start | stop) >> 1);
    if (k1 > 4u || c == TK_C_SPEC) return 0;
    const bool nxt_end = (stop >> k1) & 1u;
    uint32_t e = 0;
    if (pat.fam() == TK_PAT_O200K) {
        uint32_t ks = 64;
        if ((TK_M_WORD >> c) & 1u) ks = 0;
        else if (c != TK_C_NL && c != TK_C_NU && !nxt_end && (((w.get(TKB_UP) | w.get(TKB_LOW)) >> k1) & 1u)) ks = k1;
        if (ks != 64u) {
            const uint32_t re = ks + tk_run32(w.get(TKB_UP), stop, ks);
            if (re > TK_WIN32_SAFE) return 0;
            const uint32_t te = re + tk_run32(w.get(TKB_LOW), stop, re);
            if (te > TK_WIN32_SAFE) return 0;
            if (te > re) {
                e = te;
            } else {
                const uint32_t xx = w.get(TKB_CAS) & ~stop & tk_below32(re) & ~tk_below32(ks);
                e = xx ? 32u - tk_w32_clz(xx) : re;
            }
            if (!((stop >> e) & 1u) && a.byte(p + e) == '\'') e += tk_contraction_bits(w, a, p, e, pat);
            return e;
        }
    } else if (pat.fam() == TK_PAT_CL100K) {
        if (c == TK_C_AP) {
            const uint32_t k = tk_contraction_bits(w, a, p, 0, pat);
            if (k) return k;
        }
        if ((TK_M_L >> c) & 1u) {
            const uint32_t r = tk_run32(w.get(TKB_L), stop, 0);
            return r > TK_WIN32_SAFE ? 0 : r;
        }
        if (c != TK_C_NL && c != TK_C_NU && !nxt_end && ((w.get(TKB_L) >> k1) & 1u)) {
            const uint32_t r = k1 + tk_run32(w.get(TKB_L), stop, k1);
            return r > TK_WIN32_SAFE ? 0 : r;
        }
    } else {
        if (c == TK_C_AP) {
            const uint32_t k = tk_contraction_bits(w, a, p, 0, pat);
            if (k) return k;
        }
    }
    if (pat.fam() != TK_PAT_R50K && c == TK_C_NU) {
        const uint32_t r = tk_run32(w.get(TKB_NU), stop, 0);
        uint32_t sx = pat.digits() ? (w.start & tk_below32(r)) : 0u;
        for (uint32_t i = 0; i < pat.digits(); ++i) {
            if (pat.generic() && !sx) break;
            sx &= sx - 1;
        }
        e = sx ? tk_w32_ctz(sx) : r;
        return e > TK_WIN32_SAFE ? 0 : e;
    }
    // optional single space, then a run of one kind (r50k: letters / digits / other; others: other only)
    uint32_t s = 0;
    if (c == TK_C_SP && !nxt_end) s = k1;
    if (pat.fam() == TK_PAT_R50K) {
        int kind = -1;
        if ((w.get(TKB_L) >> s) & 1u) kind = TKB_L;
        else if ((w.get(TKB_NU) >> s) & 1u) kind = TKB_NU;
        else if ((w.get(TKB_OTH) >> s) & 1u) kind = TKB_OTH;
        if (kind >= 0) {
            const uint32_t r = s + tk_run32(w.get(kind), stop, s);
            return r > TK_WIN32_SAFE ? 0 : r;
        }
    } else if ((w.get(TKB_OTH) >> s) & 1u) {
        const uint32_t e1 = s + tk_run32(w.get(TKB_OTH), stop, s);
        if (e1 > TK_WIN32_SAFE) return 0;
        const uint32_t e2 = e1 + tk_run32((pat.fam() == TK_PAT_O200K || pat.generic()) ? w.get(TKB_NLSL) : w.get(TKB_NL), stop, e1);
        return e2 > TK_WIN32_SAFE ? 0 : e2;
    }
    // white space
    const uint32_t q = tk_run32(w.get(TKB_WS), stop, 0);
    if (q > TK_WIN32_SAFE) return 0;
    const uint32_t rng = tk_below32(q);
    const bool at_end = (stop >> q) & 1u;
    const uint32_t nlr = w.get(TKB_NL) & rng, st = w.start & rng;
    if (pat.ws_dollar() && at_end) return q;
    if (pat.nl_rule() && nlr) return 32u - tk_w32_clz(nlr);
    if (at_end) return q;
    if (tk_w32_popc(st) >= 2u) return 31u - tk_w32_clz(st);
    return q;
}

// ------------------------------------------------------------------------------------------
// table probes
// ------------------------------------------------------------------------------------------
// Exact bytes -> rank probes, one per length class (tk_common.h).  The `_from` forms continue from a slot the caller has already
// loaded (so that the first loads of several independent probes can be in flight together).
// Short: key = the 1..4 bytes packed little-endian.
TK_HD uint32_t tk_probe_short_from(const TkTables& T, uint32_t key, uint32_t len, uint32_t i, TkShortSlot s) {
    for (;;) {
        if (s.key == key && (s.val >> 30) == len - 1u && s.val != TK_SHORT_EMPTY) return s.val & TK_SHORT_MAX_RANK;
        if (s.val == TK_SHORT_EMPTY) return TK_RANK_MAX;
        i = (i + 1u) & T.short_mask;
        s = T.short_tab[i];
    }
}
TK_HD uint32_t tk_probe_short(const TkTables& T, uint32_t key, uint32_t len) {
    const uint32_t i = tk_short_slot(key, T.short_shift);
    return tk_probe_short_from(T, key, len, i, T.short_tab[i]);
}
// Mid: key = the bytes (<= 8) packed little-endian.
TK_HD uint32_t tk_probe_mid_from(const TkTables& T, uint64_t key, uint32_t len, uint32_t i, TkPieceSlot s) {
    for (;;) {
        if (s.key == key && s.len == len) return s.rank;
        if (s.len == 0u) return TK_RANK_MAX;
        i = (i + 1u) & T.mid_mask;
        s = T.mid_tab[i];
    }
}
TK_HD uint32_t tk_probe_mid(const TkTables& T, uint64_t key, uint32_t len) {
    const uint32_t i = tk_mid_slot(key, T.mid_shift);
    return tk_probe_mid_from(T, key, len, i, T.mid_tab[i]);
}
// Long (> 8 bytes): key = tk hash of the bytes; the candidate is verified byte for byte against the token blob through `verify(off)`.
template <class Verify>
TK_HD uint32_t tk_probe_piece_from(const TkTables& T, uint64_t key, uint32_t len, uint64_t i, TkPieceSlot s, Verify verify) {
    for (;;) {
        if (s.key == TK_EMPTY_KEY && s.len == 0u) return TK_RANK_MAX;
        if (s.key == key && s.len == len && verify(T.piece_off[i])) return s.rank;
        i = (i + 1) & T.piece_mask;
        s = T.piece[i];
    }
}
template <class Verify>
TK_HD uint32_t tk_probe_piece(const TkTables& T, uint64_t key, uint32_t len, Verify verify) {
    const uint64_t i = tk_piece_slot_hash(key, len) & T.piece_mask;
    return tk_probe_piece_from(T, key, len, i, T.piece[i], verify);
}

// Tokens of TK_XL_MIN..TK_XL_MAX bytes by identity (tk_common.h, tk_ident): the slot holds the bytes, a match is exact.  `s` = slot i,
// which the caller has loaded (together with whatever else it needs: one memory round trip).
TK_HD uint32_t tk_probe_xl_from(const TkTables& T, uint64_t w0, uint64_t w1, uint64_t w2, uint32_t i, TkXlSlot s) {
    for (;;) {
        if (s.rank == TK_RANK_MAX) return TK_RANK_MAX;
        if (s.w0 == w0 && s.w1 == w1 && s.w2 == w2) return s.rank;
        i = (i + 1u) & T.xl_mask;
        s = T.xl[i];
    }
}
TK_HD uint32_t tk_probe_xl(const TkTables& T, uint64_t w0, uint64_t w1, uint64_t w2) {
    const uint32_t i = (uint32_t)tk_ident_hash(w0, w1, w2, true) & T.xl_mask;
    return tk_probe_xl_from(T, w0, w1, w2, i, T.xl[i]);
}

// Pair probe.  Packed format: 4-slot (32-byte, 32-byte-aligned) buckets; a bucket is fetched with two
// 16-byte loads issued together, so a probe is one memory round trip.  A key lives in the first bucket
// of its probe sequence that had a free slot at build time, so a bucket with a free slot ends the search.
TK_HD uint32_t tk_probe_pair(const TkTables& T, uint32_t a, uint32_t b) {
    if (T.pair8) {
        const uint64_t key = ((uint64_t)a << TK_PAIR8_ID_BITS) | b;  // 42 bits
        uint64_t bk = tk_pair_slot_hash(key) & T.pair_mask;          // pair_mask counts buckets here
        for (;;) {
            const uint64_t* p = T.pair8 + bk * 4;
#if defined(__HIP_DEVICE_COMPILE__)
            const ulonglong2 v0 = *(const ulonglong2*)p, v1 = *(const ulonglong2*)(p + 2);
            const uint64_t s0 = v0.x, s1 = v0.y, s2 = v1.x, s3 = v1.y;
#else
            const uint64_t s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
#endif
            if ((s0 >> 22) == key) return (uint32_t)(s0 & 0x3FFFFFu);
            if ((s1 >> 22) == key) return (uint32_t)(s1 & 0x3FFFFFu);
            if ((s2 >> 22) == key) return (uint32_t)(s2 & 0x3FFFFFu);
            if ((s3 >> 22) == key) return (uint32_t)(s3 & 0x3FFFFFu);
            if (s3 == TK_EMPTY_KEY) return TK_RANK_MAX;  // slots fill in order, so s3 empty <=> bucket not full
            bk = (bk + 1) & T.pair_mask;
        }
    }
    uint64_t key = ((uint64_t)a << 32) | b;
    uint64_t i = tk_pair_slot_hash(key) & T.pair_mask;
    for (;;) {
        TkPairSlot s = T.pair[i];
        if (s.key == key) return s.rank;
        if (s.key == TK_EMPTY_KEY) return TK_RANK_MAX;
        i = (i + 1) & T.pair_mask;
    }
}

// 8 text bytes starting at byte offset `pos` (little-endian), from a buffer that is readable
// for 16 bytes past `pos` (callers pad).  Built from two aligned 8-byte loads.
TK_HD uint64_t tk_load8(const uint8_t* __restrict__ text, uint64_t pos) {
    uintptr_t a = (uintptr_t)(text + pos);
    const uint64_t* w = (const uint64_t*)(a & ~(uintptr_t)7);
    uint32_t sh = (uint32_t)(a & 7u) * 8u;
    uint64_t lo = w[0];
    if (sh == 0) return lo;
    uint64_t hi = w[1];
    return (lo >> sh) | (hi << (64u - sh));
}

TK_HD uint64_t tk_mask_low_bytes(uint64_t w, uint32_t nbytes) {  // keep the low nbytes (1..8)
    return nbytes >= 8u ? w : (w & ((1ull << (nbytes * 8u)) - 1ull));
}

// key of text[pos .. pos+len) (same function as host tk_key_of_bytes)
TK_HD uint64_t tk_key_of_text(const uint8_t* __restrict__ text, uint64_t pos, uint32_t len) {
    if (len <= 8u) return tk_mask_low_bytes(tk_load8(text, pos), len);
    if (len > TK_KEY_SAMPLED) {
        uint64_t h = TK_HASH_SEED ^ ((uint64_t)len << 32);
        h = tk_hash_step(h, tk_load8(text, pos));
        h = tk_hash_step(h, tk_load8(text, pos + 8u));
        h = tk_hash_step(h, tk_load8(text, pos + len - 16u));
        h = tk_hash_step(h, tk_load8(text, pos + len - 8u));
        return h == TK_EMPTY_KEY ? 0 : h;
    }
    uint64_t h = TK_HASH_SEED;
    uint32_t i = 0;
    for (; i + 8u <= len; i += 8u) h = tk_hash_step(h, tk_load8(text, pos + i));
    if (i < len) h = tk_hash_step(h, tk_mask_low_bytes(tk_load8(text, pos + i), len - i));
    if (h == TK_EMPTY_KEY) h = 0;
    return h;
}

// exact compare of text[pos..pos+len) with tok_bytes[off..off+len)
TK_HD bool tk_equal_bytes(const uint8_t* text, uint64_t pos, const uint8_t* blob, uint32_t off, uint32_t len) {
    uint32_t i = 0;
    for (; i + 8u <= len; i += 8u)
        if (tk_load8(text, pos + i) != tk_load8(blob, (uint64_t)off + i)) return false;
    if (i < len) {
        uint32_t r = len - i;
        if (tk_mask_low_bytes(tk_load8(text, pos + i), r) != tk_mask_low_bytes(tk_load8(blob, (uint64_t)off + i), r)) return false;
    }
    return true;
}

// Whole-piece probe of text[pos..pos+len)  (src/lib.rs:367)
TK_HD uint32_t tk_lookup_text_piece(const TkTables& T, const uint8_t* __restrict__ text, uint64_t pos, uint32_t len) {
    if (len <= 8u) {
        const uint64_t key = tk_mask_low_bytes(tk_load8(text, pos), len);
        if (len <= 4u && T.short_tab) return tk_probe_short(T, (uint32_t)key, len);
        return tk_probe_mid(T, key, len);
    }
    if (len > T.max_token_len) return TK_RANK_MAX;  // longer than every token
    uint64_t key = tk_key_of_text(text, pos, len);
    return tk_probe_piece(T, key, len, [&](uint32_t off) { return tk_equal_bytes(text, pos, T.tok_bytes, off, len); });
}

// ------------------------------------------------------------------------------------------
// bit helpers usable from host test builds as well
// ------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
TK_HD int tk_ffs32(uint32_t v) { return __ffs((int)v); }
TK_HD int tk_clz32(uint32_t v) { return __clz((int)v); }
TK_HD int tk_popc32(uint32_t v) { return __popc(v); }
#else
TK_HD int tk_ffs32(uint32_t v) { return __builtin_ffs((int)v); }
TK_HD int tk_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
TK_HD int tk_popc32(uint32_t v) { return __builtin_popcount(v); }
#endif

// ------------------------------------------------------------------------------------------
// byte_pair_merge (src/lib.rs:140-196) for a piece of 2..128 bytes handled by ONE lane: ids and pair
// ranks of the parts in a strided scratch (LDS: k-major, lane-minor), alive positions in a 128-bit
// register mask, so the only memory latency per merge is the pair of table probes.  Tokens (ids of
// the surviving parts, left to right) go to out[0..count).
// ------------------------------------------------------------------------------------------
struct TkMask128 {
    uint64_t lo, hi;
    TK_HD void clear(uint32_t k) {
        if (k < 64u) lo &= ~(1ull << k);
        else hi &= ~(1ull << (k - 64u));
    }
    TK_HD bool test(uint32_t k) const { return k < 64u ? (lo >> k) & 1ull : (hi >> (k - 64u)) & 1ull; }
    // lowest set position > k, or 128
    TK_HD uint32_t next_after(uint32_t k) const {
        if (k < 63u) {
            uint64_t l = lo & ~((2ull << k) - 1ull);
            if (l) return tk_ctz64(l);
        }
        uint64_t h = hi;
        if (k >= 127u) h = 0;
        else if (k >= 64u) h &= ~((2ull << (k - 64u)) - 1ull);
        if (h) return 64u + tk_ctz64(h);
        return 128u;
    }
    // highest set position < k, or -1
    TK_HD int prev_before(uint32_t k) const {
        uint64_t h = k > 64u ? hi & ((1ull << (k - 64u)) - 1ull) : 0ull;
        if (h) return 127 - (int)tk_clz64(h);
        uint64_t l = k >= 64u ? lo : lo & ((1ull << k) - 1ull);
        if (l) return 63 - (int)tk_clz64(l);
        return -1;
    }
};

template <int STRIDE>
TK_HD uint32_t tk_lane_merge(const TkTables& T, const uint8_t* __restrict__ text, uint64_t s, uint32_t n, uint32_t* id, uint32_t* rk,
                             uint32_t* __restrict__ out) {
    uint32_t pb = text[s];
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t nb = k + 1 < n ? text[s + k + 1] : 0u;
        id[k * STRIDE] = T.byte_rank[pb];
        rk[k * STRIDE] = k + 1 < n ? T.pair2[(pb << 8) | nb] : TK_RANK_MAX;
        pb = nb;
    }
    TkMask128 alive;
    alive.lo = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    alive.hi = n > 64u ? (n >= 128u ? ~0ull : ((1ull << (n - 64u)) - 1ull)) : 0ull;
    for (;;) {
        uint32_t best = TK_RANK_MAX, bi = 0;
        for (uint32_t k = 0; k + 1 < n; ++k) {
            uint32_t r = rk[k * STRIDE];
            if (r < best) {  // strict '<': leftmost minimum (lib.rs:151,190)
                best = r;
                bi = k;
            }
        }
        if (best == TK_RANK_MAX) break;
        const uint32_t j = alive.next_after(bi);  // the part being absorbed
        alive.clear(j);
        id[bi * STRIDE] = best;
        rk[j * STRIDE] = TK_RANK_MAX;
        const uint32_t nn = alive.next_after(bi);
        const int pp = alive.prev_before(bi);
        uint32_t r_i = TK_RANK_MAX, r_p = TK_RANK_MAX;
        if (nn < 128u) r_i = tk_probe_pair(T, best, id[nn * STRIDE]);
        if (pp >= 0) r_p = tk_probe_pair(T, id[pp * STRIDE], best);
        rk[bi * STRIDE] = r_i;
        if (pp >= 0) rk[pp * STRIDE] = r_p;
    }
    uint32_t t = 0;
    for (uint32_t k = 0; k < n; ++k)
        if (alive.test(k)) out[t++] = id[k * STRIDE];
    return t;
}
