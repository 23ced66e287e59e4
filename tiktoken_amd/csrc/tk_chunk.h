// Classification of the text 16 bytes per lane (host-compilable, like tk_device.h: the front kernel in tk_fused.h is the only
// product caller; the CPU-side unit tests drive the same functions sequentially).
//
// The regex classes of reference src/lib.rs:365 (\p{L} \p{N} \p{M} \s ... of the stock patterns) are a 4-bit class per
// character (tk_common.h).  Instead of one byte per lane + one ballot per class set, every lane keeps its OWN 16 text bytes
// and builds 16-bit masks (bit j = byte j of the chunk) directly in registers:
//
//   1. table pass     one LDS table lookup per byte (256 entries x {class planes, flag planes}) and two shift-or
//                     accumulations: ASCII bytes are classified, lead / continuation bytes are flagged;
//   2. decode pass    per lead byte of the chunk (and for the char that straddles in from the left): code point ->
//                     two-stage Unicode class table -> the class is OR-ed into the planes of all bytes of the char;
//   3. set algebra    class-set masks (letters, white space, ...) and the certain-start mask are boolean functions of the
//                     four planes, evaluated on 16-bit masks with full-rate VALU instructions.
//
// The measured motivation is in profiles/r02_front_phases_before.csv: the one-byte-per-lane form cost 158 vector
// instructions per 64 text bytes for these steps; this form needs about 25.
#pragma once
#include "tk_common.h"

// ---- byte table -----------------------------------------------------------------------------------------------------
// entry[b] = {planes, flags}
//   planes: bit 8p = bit p of the class nibble (ASCII bytes; 0 for bytes >= 0x80)
//   flags : bit 0 continuation byte (0x80..0xBF), bit 8 / 16 / 24 lead byte of a 2 / 3 / 4-byte char
inline void tk_build_byte_table(const uint8_t* stage1, const uint8_t* stage2, uint32_t* out /* [256 * 2] */) {
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t planes = 0, flags = 0;
        if (b < 0x80u) {
            const uint32_t c = stage2[(uint32_t)stage1[0] * 256u + b];
            for (uint32_t p = 0; p < 4; ++p) planes |= ((c >> p) & 1u) << (8u * p);
        } else if (b < 0xC0u) {
            flags = 1u;
        } else if (b < 0xE0u) {
            flags = 1u << 8;
        } else if (b < 0xF0u) {
            flags = 1u << 16;
        } else {
            flags = 1u << 24;
        }
        out[2 * b] = planes;
        out[2 * b + 1] = flags;
    }
}

struct TkChunk {
    uint32_t acc0, acc1;  // class planes of positions 0..7 / 8..15: byte p of the word = plane p, bit j = position
    uint32_t f0, f1;      // flag planes, same layout: plane 0 continuation, planes 1..3 lead byte of a 2/3/4-byte char
};

// plane p (16 bits) out of the two accumulators
TK_HD uint32_t tk_plane16(uint32_t a0, uint32_t a1, uint32_t p) { return ((a0 >> (8u * p)) & 0xFFu) | (((a1 >> (8u * p)) & 0xFFu) << 8); }

// 1. table pass.  tab(b, x, y) -> entry of byte b
template <class Tab>
TK_HD void tk_chunk_table_pass(const uint32_t w[4], Tab& tab, TkChunk& c) {
    c.acc0 = c.acc1 = c.f0 = c.f1 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; ++k) {
        const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        uint32_t x, y;
        tab(b, x, y);
        if (k < 8) {
            c.acc0 |= x << k;
            c.f0 |= y << k;
        } else {
            c.acc1 |= x << (k - 8);
            c.f1 |= y << (k - 8);
        }
    }
}

// code point of the UTF-8 sequence whose bytes are `four` (little-endian: lead byte in bits 0..7); valid UTF-8 assumed
TK_HD uint32_t tk_utf8_cp(uint32_t four, uint32_t* len_out) {
    const uint32_t b0 = four & 0xFFu;
    const uint32_t len = 2u + (uint32_t)(b0 >= 0xE0u) + (uint32_t)(b0 >= 0xF0u);
    const uint32_t x = ((b0 & 0x3Fu) << 18) | (((four >> 8) & 0x3Fu) << 12) | (((four >> 16) & 0x3Fu) << 6) | ((four >> 24) & 0x3Fu);
    const uint32_t mask = len == 2u ? 0x7FFu : (len == 3u ? 0xFFFFu : 0x1FFFFFu);
    *len_out = len;
    return (x >> (6u * (4u - len))) & mask;
}

// class c OR-ed into the planes of positions [k, k + len) that fall inside the chunk (k may be negative)
TK_HD void tk_chunk_apply(TkChunk& c, int k, uint32_t len, uint32_t cls) {
    const uint32_t ones = (1u << len) - 1u;
    const uint32_t m = (k >= 0 ? ones << k : ones >> (uint32_t)(-k)) & 0xFFFFu;
    const uint32_t pl = ((cls * 0x00204081u) & 0x01010101u) * 0xFFu;  // byte p = 0xFF iff bit p of the class
    c.acc0 |= pl & ((m & 0xFFu) * 0x01010101u);
    c.acc1 |= pl & ((m >> 8) * 0x01010101u);
}

// 2. decode pass.  get4(k) -> the four text bytes at chunk-relative offset k (-3 <= k <= 15), cls_of(cp) -> class nibble.
// prev: the four bytes before the chunk (byte 3 of `prev` = position -1) locate the lead of a char that straddles in from the
// previous chunk (has_prev: those bytes exist).
// TWO chars per step (round 6): a step is a chain of dependent table loads, and a wavefront takes as many steps as its lane with the
// most chars (a chunk of CJK text has five or six; the average lane of web text less than two) -- with two chars per step the loads of
// both are in flight together and the steps are half as many.  A lane with an odd number applies its last char twice (an OR: harmless).
template <class Get4, class ClsOf>
TK_HD void tk_chunk_decode(TkChunk& c, uint32_t prev, bool has_prev, Get4& get4, ClsOf& cls_of) {
    const uint32_t cont = tk_plane16(c.f0, c.f1, 0);
    uint32_t leads = tk_plane16(c.f0, c.f1, 1) | tk_plane16(c.f0, c.f1, 2) | tk_plane16(c.f0, c.f1, 3);
    bool pend = (cont & 1u) && has_prev;  // the char that straddles in: its lead is one to three bytes before the chunk
    const int kprev = (prev >> 24) >= 0xC0u ? -1 : (((prev >> 16) & 0xFFu) >= 0xC0u ? -2 : -3);
    auto next_lead = [&]() -> int {
#if defined(__HIP_DEVICE_COMPILE__)
        const int k = __ffs((int)leads) - 1;
#else
        const int k = __builtin_ctz(leads);
#endif
        leads &= leads - 1;
        return k;
    };
    while (pend || leads) {
        int k1;
        if (pend) {
            k1 = kprev;
            pend = false;
        } else {
            k1 = next_lead();
        }
        const int k2 = leads ? next_lead() : k1;
        uint32_t len1, len2;
        const uint32_t cp1 = tk_utf8_cp(get4(k1), &len1), cp2 = tk_utf8_cp(get4(k2), &len2);
        const uint32_t c1 = cls_of(cp1), c2 = cls_of(cp2);
        tk_chunk_apply(c, k1, len1, c1);
        tk_chunk_apply(c, k2, len2, c2);
    }
}

// Final masks of a chunk.  valid / past: positions inside the text / at or after its end; brk: break bitmap bits (document
// starts, special-token edges); ss / si: first / interior bytes of allowed special-token occurrences (0 without specials).
struct TkChunkMasks {
    uint32_t p[4];               // class planes (continuation bytes carry their char's class)
    uint32_t start, hard, text;  // char starts; hard starts (look-ahead sees end-of-text there); char starts of real text
};
TK_HD void tk_chunk_finalize(const TkChunk& c, uint32_t valid, uint32_t past, uint32_t brk, uint32_t ss, uint32_t si, TkChunkMasks& m) {
    const uint32_t inv = ~valid & 0xFFFFu;
    uint32_t cont = tk_plane16(c.f0, c.f1, 0) & valid;
    // positions outside the text are class END (1100b); special-token bytes are class SPEC (1101b), their interior is "continuation"
    const uint32_t sm = (ss | si) & valid;
    m.p[0] = (tk_plane16(c.acc0, c.acc1, 0) & valid) | sm;
    m.p[1] = tk_plane16(c.acc0, c.acc1, 1) & valid & ~sm;
    m.p[2] = tk_plane16(c.acc0, c.acc1, 2) | inv | sm;
    m.p[3] = tk_plane16(c.acc0, c.acc1, 3) | inv | sm;
    cont = (cont & ~ss) | (si & valid);
    m.start = ~cont & 0xFFFFu;
    m.text = m.start & valid;
    m.hard = (brk & m.text) | (past & 0xFFFFu) | (ss & ~si & valid);
}

// class nibble of position j from the four planes
TK_HD uint32_t tk_class_from_planes(const uint32_t p[4], uint32_t j) {
    return ((p[0] >> j) & 1u) | (((p[1] >> j) & 1u) << 1) | (((p[2] >> j) & 1u) << 2) | (((p[3] >> j) & 1u) << 3);
}

// 3. set algebra on 16-bit masks.  P[p] = plane p; classes are tk_common.h's TK_C_* codes.
struct TkSets {
    uint32_t ws, nl, sp, wso, l, lu, ll, lc, mk, nu, ap, sl, ot, oth, word, up, low, cas, nlsl, end, spec;
};
TK_HD void tk_sets_from_planes(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, TkSets& s) {
    const uint32_t m = 0xFFFFu;
    const uint32_t n0 = ~p0 & m, n1 = ~p1 & m, n2 = ~p2 & m, n3 = ~p3 & m;
    const uint32_t q00 = n3 & n2, q01 = n3 & p2, q10 = p3 & n2, q11 = p3 & p2;  // class >> 2
    const uint32_t r00 = n1 & n0, r01 = n1 & p0, r10 = p1 & n0, r11 = p1 & p0;  // class & 3
    s.nl = q00 & r01;
    s.sp = q00 & r10;
    s.wso = q00 & r11;
    s.ws = q00 & ~r00;
    s.lu = q01 & r00;
    s.ll = q01 & r01;
    s.lc = q01 & r10;
    s.mk = q01 & r11;
    s.l = q01 & ~r11;
    s.word = q01;                 // L | MK
    s.up = q01 & ~r01;            // LU LC MK
    s.low = q01 & ~r00;           // LL LC MK
    s.cas = q01 & p1;             // LC MK
    s.nu = q10 & r00;
    s.ap = q10 & r01;
    s.sl = q10 & r10;
    s.ot = q10 & r11;
    s.oth = s.mk | (q10 & ~r00);  // MK AP SL OT
    s.nlsl = s.nl | s.sl;
    s.end = q11 & r00;
    s.spec = q11 & r01;
}

// `set` of the PREVIOUS byte: shifted by one position, bit 0 taken from the class of the byte before the chunk
TK_HD uint32_t tk_prev_set(uint32_t set, uint32_t class_mask, uint32_t prevc) { return ((set << 1) & 0xFFFFu) | ((class_mask >> prevc) & 1u); }

// Certain piece starts of the chunk: char starts where a boundary is certain whatever the left context (tk_device.h
// tk_certain_mask restated on masks).  prevc = class nibble of the byte before the chunk (0 when unknown: never certain).
// near: positions with an apostrophe two or three bytes before them (tk_chunk_near) -- there a lower-case letter followed by an upper-case
// one may be the inside of a contraction ("'lL"); everywhere else it is a boundary of the o200k pattern ("camelCase").
TK_HD uint32_t tk_chunk_certain(int pat, const TkSets& s, uint32_t start, uint32_t hard, uint32_t prevc, uint32_t near) {
    const uint32_t L3 = TK_M_L, O4 = TK_M_OTHER, NU = TK_CB(TK_C_NU);
    uint32_t cert = hard;
    if (pat == TK_PAT_R50K) {
        cert |= tk_prev_set(s.nl | s.wso, TK_CB(TK_C_NL) | TK_CB(TK_C_WSO), prevc) & (s.l | s.oth | s.nu);
        cert |= tk_prev_set(s.l, L3, prevc) & (s.ws | s.oth | s.nu);
        cert |= tk_prev_set(s.mk | s.sl | s.ot, TK_CB(TK_C_MK) | TK_CB(TK_C_SL) | TK_CB(TK_C_OT), prevc) & (s.ws | s.l | s.nu);
        cert |= tk_prev_set(s.ap, TK_CB(TK_C_AP), prevc) & (s.ws | s.lu | s.lc | s.nu);
        cert |= tk_prev_set(s.nu, NU, prevc) & (s.ws | s.l | s.oth);
    } else if (pat == TK_PAT_CL100K) {
        cert |= tk_prev_set(s.nl, TK_CB(TK_C_NL), prevc) & (s.l | s.oth | s.nu);
        cert |= tk_prev_set(s.sp, TK_CB(TK_C_SP), prevc) & s.nu;
        cert |= tk_prev_set(s.wso, TK_CB(TK_C_WSO), prevc) & (s.oth | s.nu);
        cert |= tk_prev_set(s.l, L3, prevc) & (s.ws | s.oth | s.nu);
        cert |= tk_prev_set(s.oth, O4, prevc) & (s.sp | s.wso | s.nu);
        cert |= tk_prev_set(s.nu, NU, prevc) & (s.ws | s.l | s.oth);
    } else {
        cert |= tk_prev_set(s.nl, TK_CB(TK_C_NL), prevc) & (s.l | s.mk | s.nu | s.ap | s.ot);
        cert |= tk_prev_set(s.sp, TK_CB(TK_C_SP), prevc) & s.nu;
        cert |= tk_prev_set(s.wso, TK_CB(TK_C_WSO), prevc) & (s.nu | s.ap | s.sl | s.ot);
        cert |= tk_prev_set(s.l, L3, prevc) & (s.ws | s.nu | s.sl | s.ot);
        cert |= tk_prev_set(s.ll, TK_CB(TK_C_LL), prevc) & s.lu & ~near;
        cert |= tk_prev_set(s.oth, O4, prevc) & (s.sp | s.wso | s.nu);
        cert |= tk_prev_set(s.nu, NU, prevc) & (s.ws | s.l | s.oth);
    }
    return cert & start;  // (pass the char starts of REAL text: positions past the end are hard but never pieces)
}

// Positions of the chunk with an apostrophe two or three bytes before them: the end of a contraction ('s: apostrophe + 2 bytes; 'll,
// 'ſ: + 3).  ap = the chunk's apostrophe bytes; ap_before: bits 0..2 = apostrophe at byte -3, -2, -1 before the chunk (7 when unknown).
TK_HD uint32_t tk_chunk_near(uint32_t ap, uint32_t ap_before) {
    return ((ap << 2) | (ap << 3) | ((ap_before & 1u) ? 1u : 0u) | ((ap_before & 2u) ? 3u : 0u) | ((ap_before & 4u) ? 6u : 0u)) & 0xFFFFu;
}

// Char starts of the chunk at which NO piece of the stock pattern `pat` can start: the class pair (previous char, this char) is in
// tk_never_mask (tk_device.h) and the position is not `near` an apostrophe (tk_chunk_near).  The rows are written out as set algebra;
// the unit tests check them against the table.
TK_HD uint32_t tk_chunk_never(int pat, const TkSets& s, uint32_t prevc, uint32_t near) {
    const uint32_t L3 = TK_M_L, O4 = TK_M_OTHER;
    uint32_t nev = 0;
    if (pat == TK_PAT_R50K) {
        nev |= tk_prev_set(s.sp, TK_CB(TK_C_SP), prevc) & (s.l | s.oth | s.nu);
        nev |= tk_prev_set(s.l, L3, prevc) & s.l;
        nev |= tk_prev_set(s.nu, TK_CB(TK_C_NU), prevc) & s.nu;
        nev |= tk_prev_set(s.oth, O4, prevc) & s.oth;
    } else if (pat == TK_PAT_CL100K) {
        nev |= tk_prev_set(s.nl, TK_CB(TK_C_NL), prevc) & s.nl;
        nev |= tk_prev_set(s.sp, TK_CB(TK_C_SP), prevc) & (s.nl | s.l | s.oth);
        nev |= tk_prev_set(s.wso, TK_CB(TK_C_WSO), prevc) & (s.nl | s.l);
        nev |= tk_prev_set(s.l, L3, prevc) & s.l;
        nev |= tk_prev_set(s.oth, O4, prevc) & (s.nl | s.oth);
    } else {
        nev |= tk_prev_set(s.nl, TK_CB(TK_C_NL), prevc) & s.nl;
        nev |= tk_prev_set(s.sp, TK_CB(TK_C_SP), prevc) & (s.nl | s.l | s.oth);
        nev |= tk_prev_set(s.wso, TK_CB(TK_C_WSO), prevc) & (s.nl | s.l | s.mk);
        nev |= tk_prev_set(s.lu, TK_CB(TK_C_LU), prevc) & (s.l | s.mk);
        nev |= tk_prev_set(s.ll | s.lc, TK_CB(TK_C_LL) | TK_CB(TK_C_LC), prevc) & (s.ll | s.lc | s.mk);
        nev |= tk_prev_set(s.mk, TK_CB(TK_C_MK), prevc) & s.mk;
        nev |= tk_prev_set(s.ap | s.ot, TK_CB(TK_C_AP) | TK_CB(TK_C_OT), prevc) & (s.nl | s.oth);
        nev |= tk_prev_set(s.sl, TK_CB(TK_C_SL), prevc) & (s.nl | s.sl);
    }
    return nev & ~near;  // (the end of a contraction is a boundary whatever the classes say)
}

// The "second stop" rule (round 6).  A piece that starts at a certain start ends at the next stop when that stop is certain (the scanners'
// short cut, tk_fused.h phase D).  When it is not, one case is still decided without a scanner: the start is a one-byte char of a class
// the pattern's letter alternative takes as its optional PREFIX (`[^\r\n\p{L}\p{N}]?` of the cl100k / o200k patterns), and the stop is
// the very next byte, a letter.  Then the piece is the prefix char plus at least one letter (an alternative with the prefix matches
// whenever a letter follows), so it does not end at that stop: it ends at the stop BEHIND it -- if that one is certain; every piece end is
// a stop, and there is no other in between.  41 % of the starts the scanners were given on web text (".com", "(x", "\"hello").
//   o200k : prefix classes SP WSO AP SL OT (no alternative of its own for "'s": the contraction is a suffix), letters LU LL LC MK
//   cl100k: the same without AP (an apostrophe may start the contraction alternative, which comes first), letters LU LL LC
//   r50k  : none (its only prefix is a space, and space -> letter is never a stop)
// The unit tests check the rule against the sequential scanner on adversarial text (tests/test_device_logic_sim.py).
TK_HD uint32_t tk_second_stop_prefix_classes(int pat) {
    const uint32_t x = TK_CB(TK_C_SP) | TK_CB(TK_C_WSO) | TK_CB(TK_C_SL) | TK_CB(TK_C_OT);
    return pat == TK_PAT_O200K ? (x | TK_CB(TK_C_AP)) : (pat == TK_PAT_CL100K ? x : 0u);
}
TK_HD uint32_t tk_second_stop_letter_classes(int pat) {
    return pat == TK_PAT_O200K ? (TK_M_L | TK_CB(TK_C_MK)) : (pat == TK_PAT_CL100K ? TK_M_L : 0u);
}
// positions of the chunk that are such a stop: uncertain (`stop_unc`), a letter, and the byte before it a certain start of a prefix class
// that is a whole char (position 0 of a chunk never qualifies: its predecessor is another lane's)
TK_HD uint32_t tk_chunk_second_stop(int pat, const TkSets& s, uint32_t start, uint32_t cert, uint32_t stop_unc) {
    if (pat == TK_PAT_R50K) return 0u;
    const uint32_t prefix = pat == TK_PAT_O200K ? (s.sp | s.wso | s.ap | s.sl | s.ot) : (s.sp | s.wso | s.sl | s.ot);
    const uint32_t letter = pat == TK_PAT_O200K ? (s.l | s.mk) : s.l;
    return ((cert & prefix & start) << 1) & stop_unc & letter & start & 0xFFFFu;
}

// The same from a table given at run time (generic patterns: cert[a] = class mask, TkTables::cert)
TK_HD uint32_t tk_chunk_certain_rt(const uint16_t* cm, const TkSets& s, uint32_t start, uint32_t hard, uint32_t prevc) {
    const uint32_t set_of[12] = {0u, s.nl, s.sp, s.wso, s.lu, s.ll, s.lc, s.mk, s.nu, s.ap, s.sl, s.ot};
    uint32_t cert = hard;
    for (uint32_t a = TK_C_NL; a <= (uint32_t)TK_C_OT; ++a) {
        const uint32_t m = cm[a];
        uint32_t follow = 0;
        for (uint32_t b = TK_C_NL; b <= (uint32_t)TK_C_OT; ++b) follow |= ((m >> b) & 1u) ? set_of[b] : 0u;
        cert |= tk_prev_set(set_of[a], 1u << a, prevc) & follow;
    }
    return cert & start;
}
