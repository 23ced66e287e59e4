// CPU-side simulation of the device logic, for `-m "not gpu"` unit tests ONLY.
//
// It compiles the product's host/device header (tiktoken_amd/csrc/tk_device.h) with g++ and runs
// the same functions the kernels run -- class bytes, certain-start segmentation, tk_piece_end,
// table probes, the per-lane merge -- in a sequential loop that mirrors tk_k_front and the merge kernels.
// It lets logic errors surface in the container (no GPU here).  It is NOT a product path: the
// library never links it, and it does not replace the GPU parity tests.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

static uint64_t g_rx_matches = 0, g_rx_steps = 0;
#define TK_RX_ON_MATCH() (++g_rx_matches)
#define TK_RX_ON_DONE(steps) (g_rx_steps += (steps))
#include "../../tiktoken_amd/csrc/tk_chunk.h"
#include "../../tiktoken_amd/csrc/tk_mid_plan.h"
#include "../../tiktoken_amd/csrc/tk_regex_host.h"
#include "../../tiktoken_amd/csrc/tk_regex_split.h"
#include "../../tiktoken_amd/csrc/tk_device.h"
#include "../../tiktoken_amd/csrc/tk_tables.h"
#include "../../tiktoken_amd/csrc/tk_unicode_tables.inc"

struct Sim {
    TkHostTables H;
    TkTables T;
    std::vector<uint32_t> byte_tab;
};

struct FlatAcc {
    const uint8_t* cls_;
    const uint8_t* text;
    uint64_t n;
    uint32_t cls(uint64_t pos) const { return pos >= n ? (uint32_t)TK_C_END : cls_[pos]; }
    uint32_t byte(uint64_t pos) const { return text[pos]; }
    // run queries of tk_piece_end_runs, answered by plain loops (the kernel answers them with the whole workgroup)
    uint64_t run_end(uint64_t from, uint32_t mask) const {
        uint64_t s = from;
        while ((mask >> tk_la(*this, s)) & 1u) s = tk_next_char(*this, s);
        return s;
    }
    uint64_t last_in(uint64_t from, uint64_t to, uint32_t mask) const {
        uint64_t best = TK_NO_POS;
        for (uint64_t i = from; i < to && i < n; ++i) {
            const uint32_t c = cls_[i] & 15u;
            if (c != TK_C_CONT && ((mask >> c) & 1u)) best = i;
        }
        return best;
    }
};

static uint64_t g_runs_mismatch = 0, g_never_viol = 0;

// certain piece starts from the table that travels with the pattern (TkTables::cert: the family's static table for stock patterns)
static inline bool tk_certain_rt(const uint16_t* cert, uint32_t a, uint32_t b) { return (cert[a & 15u] >> b) & 1u; }
// ... plus the one rule with context: in the stock o200k pattern a lower-case letter followed by an upper-case one is a boundary unless an
// apostrophe stands two or three bytes before the upper-case letter ("'lL" is a contraction) -- tk_chunk_certain's `near` term
static inline bool tk_certain_ctx(const TkTables& T, uint32_t a, uint32_t b, const uint8_t* text, uint64_t i) {
    if ((T.cert[a & 15u] >> b) & 1u) return true;
    if (T.pat.generic() || T.pattern != TK_PAT_O200K || a != TK_C_LL || b != TK_C_LU) return false;
    return !((i >= 2 && text[i - 2] == '\'') || (i >= 3 && text[i - 3] == '\''));
}

extern "C" {
uint64_t tks_runs_mismatches() { return g_runs_mismatch; }
// piece starts found at a class pair that tk_never_mask calls impossible (stock patterns)
uint64_t tks_never_violations() { return g_never_viol; }

void* tks_create(const uint8_t* ranks_blob, const uint64_t* ranks_off, const uint32_t* ranks_ids, uint64_t n_ranks,
                 const uint8_t* spec_blob, const uint64_t* spec_off, const uint32_t* spec_ids, uint64_t n_spec,
                 const char* pat_str, char* err, uint64_t errcap) {
    Sim* s = new Sim();
    std::string e = tk_build_tables(ranks_blob, ranks_off, ranks_ids, n_ranks, spec_blob, spec_off, spec_ids, n_spec, pat_str, &s->H);
    if (!e.empty()) {
        strncpy(err, e.c_str(), errcap - 1);
        err[errcap - 1] = 0;
        delete s;
        return nullptr;
    }
    TkHostTables& H = s->H;
    TkTables& D = s->T;
    // (a pat_str of the generic engine: every char a letter, as tk_create sets it up -- the split is tks_rx_split's)
    static const std::vector<uint8_t> ll_stage1(0x1100, 0), ll_stage2(256, (uint8_t)TK_C_LL);
    const bool gen = !H.rx.empty();
    D.uc_stage1 = gen ? ll_stage1.data() : tk_uc_stage1;
    D.uc_stage2 = gen ? ll_stage2.data() : tk_uc_stage2;
    s->byte_tab.resize(256 * 2);
    tk_build_byte_table(D.uc_stage1, D.uc_stage2, s->byte_tab.data());
    D.byte_tab = s->byte_tab.data();
    D.short_tab = H.short_tab.empty() ? nullptr : H.short_tab.data();
    D.short_mask = H.short_mask;
    D.short_shift = H.short_shift;
    D.mid_tab = H.mid_tab.data();
    D.mid_mask = H.mid_mask;
    D.mid_shift = H.mid_shift;
    D.piece = H.piece.data();
    D.piece_off = H.piece_off.data();
    D.piece_mask = H.piece_mask;
    D.xl = H.xl.data();
    D.xl_mask = H.xl_mask;
    D.xfilter = H.xfilter.data();
    D.max_token_len = H.max_token_len;
    D.tok_bytes = H.tok_bytes.data();
    D.pair = H.pair8.empty() ? H.pair.data() : nullptr;
    D.pair8 = H.pair8.empty() ? nullptr : H.pair8.data();
    D.pair_mask = H.pair_mask;
    D.pair2 = H.pair2.data();
    D.byte_rank = H.byte_rank;
    D.spec_bytes = H.spec_bytes.data();
    D.spec_off = H.spec_off.data();
    D.spec_id = H.spec_id.data();
    D.n_spec = (uint32_t)H.spec_id.size();
    memcpy(D.spec_first, H.spec_first, sizeof D.spec_first);
    D.pattern = H.pattern;
    D.pat = H.pat;
    memcpy(D.cert, H.cert, sizeof D.cert);
    return s;
}
void tks_destroy(void* p) { delete (Sim*)p; }
uint64_t tks_n_pairs(void* p) { return ((Sim*)p)->H.n_pairs; }
// FNV-1a over every table the device gets (the build must not depend on the number of host threads)
uint64_t tks_tables_digest(void* p) {
    const TkHostTables& H = ((Sim*)p)->H;
    uint64_t h = 0xCBF29CE484222325ull;
    auto mix = [&](const void* d, size_t n) {
        const uint8_t* b = (const uint8_t*)d;
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    };
    mix(H.short_tab.data(), H.short_tab.size() * sizeof(TkShortSlot));
    for (const TkPieceSlot& e : H.mid_tab) { mix(&e.key, 8); mix(&e.rank, 4); mix(&e.len, 4); }
    for (const TkPieceSlot& e : H.piece) { mix(&e.key, 8); mix(&e.rank, 4); mix(&e.len, 4); }
    mix(H.piece_off.data(), H.piece_off.size() * 4);
    for (const TkXlSlot& e : H.xl) { mix(&e.w0, 8); mix(&e.w1, 8); mix(&e.w2, 8); mix(&e.rank, 4); }
    mix(H.xfilter.data(), H.xfilter.size() * 4);
    mix(H.pair8.data(), H.pair8.size() * 8);
    for (const TkPairSlot& e : H.pair) { mix(&e.key, 8); mix(&e.rank, 4); }
    mix(H.pair2.data(), H.pair2.size() * 4);
    mix(H.byte_rank, sizeof H.byte_rank);
    mix(H.tok_bytes.data(), H.tok_bytes.size());
    const std::vector<uint32_t>& sr = H.sorted_ranks();
    mix(sr.data(), sr.size() * 4);
    return h;
}
// table build statistics: average slots inspected per stored token {short, mid, long}, table sizes in slots
void tks_table_stats(void* p, double* probes, uint64_t* slots) {
    const TkHostTables& H = ((Sim*)p)->H;
    probes[0] = H.probes_short; probes[1] = H.probes_mid; probes[2] = H.probes_long;
    slots[0] = H.short_tab.size(); slots[1] = H.mid_tab.size(); slots[2] = H.piece.size();
}
// whole-piece lookup through the device probe functions (TK_RANK_MAX = absent)
uint32_t tks_lookup(void* p, const uint8_t* piece, uint32_t len) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> text(piece, piece + len);
    text.resize(len + 64, 0);
    return tk_lookup_text_piece(s->T, text.data(), 0, len);
}

// The front kernel's lookup of a piece of TK_XL_MIN..TK_XL_MAX bytes: its identity (tk_ident, from eight-byte loads of a text in which
// `fill` bytes FOLLOW the piece -- whatever stands there must not matter), then the identity table.  Also returns the identity and its hash.
uint32_t tks_lookup_xl(void* p, const uint8_t* piece, uint32_t len, uint8_t fill, uint64_t* ident4) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> text(piece, piece + len);
    text.resize(len + 64, fill);
    uint64_t w0, w1, w2;
    tk_ident([&](uint32_t o) { return tk_load8(text.data(), o); }, len, 0u, w0, w1, w2);
    if (ident4) {
        uint64_t hh = tk_ident_hash(w0, w1, w2, len <= TK_XL_MAX);
        ident4[0] = w0; ident4[1] = w1; ident4[2] = w2; ident4[3] = hh;
        if (hh == TK_EMPTY_KEY) hh = 0;
        const uint32_t bit = tk_xfilter_bit(hh);
        ident4[4] = (s->T.xfilter[bit >> 5] >> (bit & 31u)) & 1u;  // (means something for pieces of more than TK_XL_MAX bytes)
    }
    if (len < TK_XL_MIN || len > TK_XL_MAX) return TK_RANK_MAX;
    return tk_probe_xl(s->T, w0, w1, w2);
}

// Mirror of the byte-walking scanner (the fallback of tk_k_front): class bytes, certain starts, scanner from each certain start.
// doc_off marks hard starts.  Writes a byte per position: 1 = piece start.  Returns the number of
// certain starts (for statistics).
uint64_t tks_pretok(void* p, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, uint8_t* starts) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> text(text_in, text_in + n);
    text.resize(n + 64, 0);
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) brk[doc_off[d] >> 5] |= 1u << (doc_off[d] & 31);
    std::vector<uint8_t> cls(n + 8, TK_C_END);
    for (uint64_t i = 0; i < n; ++i) cls[i] = (uint8_t)tk_class_byte(s->T, text.data(), i, n, brk.data(), nullptr, nullptr);
    memset(starts, 0, n);
    FlatAcc acc{cls.data(), text.data(), n};
    const int pat = s->T.pattern;
    const uint16_t* s_cert = s->T.cert;
    (void)pat;
    const TkPat patx = s->T.pat;
    (void)patx;
    uint64_t n_certain = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = cls[i];
        if ((c & 15u) == TK_C_CONT) continue;
        bool certain = (c & TK_F_HARD) != 0;
        if (!certain) {
            if (i == 0) continue;  // the kernel sees TK_C_END to the left of position 0
            uint64_t j = i - 1;
            while (j > 0 && cls[j] == TK_C_CONT) --j;
            certain = tk_certain_ctx(s->T, cls[j] & 15u, c & 15u, text.data(), i);
        }
        if (!certain) continue;
        ++n_certain;
        starts[i] = 1;
        uint64_t q = i;
        for (;;) {
            uint64_t e = tk_piece_end(acc, q, patx);
            if (tk_piece_end_runs(acc, q, patx) != e) ++g_runs_mismatch;  // the run-query form must agree everywhere
            if (e <= q) e = tk_next_char(acc, q);
            if (e >= n) break;
            uint32_t ce = acc.cls(e);
            if (ce & TK_F_HARD) break;
            uint64_t j = e - 1;
            while (acc.cls(j) == TK_C_CONT) --j;
            if (tk_certain_ctx(s->T, acc.cls(j) & 15u, ce & 15u, text.data(), e)) break;
            starts[e] = 1;
            q = e;
        }
    }
    if (!patx.generic())  // every piece start against the table of impossible boundaries
        for (uint64_t i = 1; i < n; ++i) {
            if (!starts[i] || (cls[i] & TK_F_HARD)) continue;
            uint64_t j = i - 1;
            while (j > 0 && cls[j] == TK_C_CONT) --j;
            const bool near = (i >= 2 && text[i - 2] == '\'') || (i >= 3 && text[i - 3] == '\'');
            if (!near && ((tk_never_mask(pat, cls[j] & 15u) >> (cls[i] & 15u)) & 1u)) ++g_never_viol;
        }
    return n_certain;
}

// Mirror of the bit-parallel scanners of tk_k_front: class bytes with the char's class propagated onto
// continuation bytes, per-class bitmaps, 64-bit windows at each piece start, tk_piece_len_bits with the
// byte-walking scanner as the fallback.  Returns the number of pieces that needed the fallback.
struct PropAcc {  // accessor over the propagated class array (0x40 = continuation byte)
    const uint8_t* cls2;
    const uint8_t* text;
    uint64_t n;
    uint32_t cls(uint64_t pos) const {
        if (pos >= n) return (uint32_t)TK_C_END;
        uint32_t c = cls2[pos];
        return (c & 0x40u) ? (uint32_t)TK_C_CONT : (c & 0x8Fu);
    }
    uint32_t byte(uint64_t pos) const { return text[pos]; }
};

struct SimExt {  // extension windows from the byte-per-bit arrays; `lim` mimics the LDS window edge of a tile
    const std::vector<uint8_t>* bm[TKB_KINDS];
    uint64_t p;
    uint32_t lim;
    uint64_t win(int kind, uint32_t j) const {
        uint64_t w = 0;
        const std::vector<uint8_t>& v = *bm[kind];
        for (uint32_t k = 0; k < 64; ++k) {
            uint64_t pos = p + 64ull * j + k;
            if (pos < v.size() ? v[pos] : (kind == TKB_HARD || kind == TKB_START)) w |= 1ull << k;
        }
        return w;
    }
    uint32_t limit() const { return lim; }
};

static uint64_t win_of(const std::vector<uint8_t>& bm, uint64_t p) {
    uint64_t w = 0;
    for (uint32_t k = 0; k < 64; ++k)
        if (p + k < bm.size() && bm[p + k]) w |= 1ull << k;
    return w;
}

uint64_t tks_pretok_bits(void* pv, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, uint8_t* starts) {
    Sim* s = (Sim*)pv;
    std::vector<uint8_t> text(text_in, text_in + n);
    text.resize(n + 64, 0);
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) brk[doc_off[d] >> 5] |= 1u << (doc_off[d] & 31);
    std::vector<uint8_t> cls2(n + 80, TK_C_END | 0x80);
    uint8_t last = TK_C_OT;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = tk_class_byte(s->T, text.data(), i, n, brk.data(), nullptr, nullptr);
        if ((c & 15u) == TK_C_CONT) cls2[i] = (uint8_t)(last | 0x40);
        else {
            cls2[i] = (uint8_t)c;
            last = (uint8_t)(c & 15u);
        }
    }
    // bitmaps (one byte per bit here; the kernel packs them with __ballot)
    const TkPat patx = s->T.pat;
    const size_t m = n + 80;
    std::vector<uint8_t> b_start(m, 0), b_hard(m, 0), b_L(m, 0), b_up(m, 0), b_low(m, 0), b_cas(m, 0), b_oth(m, 0), b_ws(m, 0),
        b_nl(m, 0), b_nu(m, 0), b_nlsl(m, 0);
    for (uint64_t i = 0; i < m; ++i) {
        if (i >= n) {
            b_start[i] = 1;
            b_hard[i] = 1;
            continue;
        }
        uint32_t c = cls2[i], k = c & 15u;
        b_start[i] = !(c & 0x40u);
        b_hard[i] = (c & 0x80u) != 0;
        b_L[i] = (TK_M_L >> k) & 1u;
        b_up[i] = (TK_M_UPPERISH >> k) & 1u;
        b_low[i] = (TK_M_LOWERISH >> k) & 1u;
        b_cas[i] = k == TK_C_LC || k == TK_C_MK;
        b_oth[i] = (TK_M_OTHER >> k) & 1u;
        b_ws[i] = (TK_M_WS >> k) & 1u;
        b_nl[i] = k == TK_C_NL;
        b_nu[i] = k == TK_C_NU;
        // (generic patterns keep their suffix set in this bitmap, as the front kernel does)
        b_nlsl[i] = patx.generic() ? (((patx.suffix() & 1u) && k == TK_C_NL) || ((patx.suffix() & 2u) && k == TK_C_SL)) : (k == TK_C_NL || k == TK_C_SL);
    }
    memset(starts, 0, n);
    PropAcc acc{cls2.data(), text.data(), n};
    const int pat = s->T.pattern;
    const uint16_t* s_cert = s->T.cert;
    (void)pat;
    uint64_t n_fallback = 0, n_fast32 = 0;
    (void)n_fast32;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = cls2[i];
        if (c & 0x40u) continue;
        bool certain = (c & 0x80u) != 0;
        if (!certain) {
            if (i == 0) continue;
            certain = tk_certain_ctx(s->T, cls2[i - 1] & 15u, c & 15u, text.data(), i);  // propagated class: no walking back
        }
        if (!certain) continue;
        starts[i] = 1;
        uint64_t q = i;
        for (;;) {
            TkWin w;
            w.start = win_of(b_start, q);
            w.stop = win_of(b_hard, q) & ~1ull;
            w.L = win_of(b_L, q);
            w.up = win_of(b_up, q);
            w.low = win_of(b_low, q);
            w.cas = win_of(b_cas, q);
            w.oth = win_of(b_oth, q);
            w.ws = win_of(b_ws, q);
            w.nl = win_of(b_nl, q);
            w.nu = win_of(b_nu, q);
            w.nlsl = win_of(b_nlsl, q);
            SimExt ext;
            const std::vector<uint8_t>* arr[TKB_KINDS] = {&b_start, &b_hard, &b_L, &b_up, &b_low, &b_cas, &b_oth, &b_ws, &b_nl, &b_nu, &b_nlsl};
            for (int kk = 0; kk < TKB_KINDS; ++kk) ext.bm[kk] = arr[kk];
            ext.p = q;
            ext.lim = (uint32_t)(4096 + 192 - (q % 4096));  // like tk_k_front: tile + look-ahead
            struct W32 {  // the kernel's 32-bit fast path sees the low halves of the same windows
                const TkWin* w;
                uint32_t start, stop;
                uint32_t get(int kind) const { return (uint32_t)w->get(kind); }
            } w32{&w, (uint32_t)w.start, (uint32_t)w.stop};
            uint32_t len = tk_piece_len_bits32(w32, acc, q, cls2[q] & 15u, patx);
            if (len) ++n_fast32;
            else len = tk_piece_len_bits(w, acc, ext, q, cls2[q] & 15u, patx);
            uint64_t e;
            if (len) {
                e = q + len;
            } else {
                ++n_fallback;
                e = tk_piece_end(acc, q, patx);
            }
            if (e <= q) e = tk_next_char(acc, q);
            if (e >= n) break;
            uint32_t ce = cls2[e];
            if (ce & 0x80u) break;
            if (tk_certain_ctx(s->T, cls2[e - 1] & 15u, ce & 15u, text.data(), e)) break;
            starts[e] = 1;
            q = e;
        }
    }
    return n_fallback;
}

// Mirror of the tile rule of tk_k_front: every tile derives the piece starts INSIDE ITS OWN byte range and
// nothing else.  Scanners start at the certain starts of the tile plus the last certain start at or
// before the tile (found in the 64-byte left context, else by walking back); a scanner only records
// boundaries that fall inside the tile and stops at the tile end.  No state crosses tiles.
uint64_t tks_pretok_tiles(void* pv, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, uint8_t* starts,
                          uint32_t tile, uint32_t left) {
    Sim* s = (Sim*)pv;
    std::vector<uint8_t> text(text_in, text_in + n);
    text.resize(n + 64, 0);
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) brk[doc_off[d] >> 5] |= 1u << (doc_off[d] & 31);
    std::vector<uint8_t> cls2(n + 80, TK_C_END | 0x80);
    uint8_t last = TK_C_OT;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = tk_class_byte(s->T, text.data(), i, n, brk.data(), nullptr, nullptr);
        if ((c & 15u) == TK_C_CONT) cls2[i] = (uint8_t)(last | 0x40);
        else {
            cls2[i] = (uint8_t)c;
            last = (uint8_t)(c & 15u);
        }
    }
    memset(starts, 0, n);
    PropAcc acc{cls2.data(), text.data(), n};
    const int pat = s->T.pattern;
    const uint16_t* s_cert = s->T.cert;
    (void)pat;
    const TkPat patx = s->T.pat;
    (void)patx;
    auto certain_at = [&](uint64_t i) -> bool {
        uint32_t c = cls2[i];
        if (c & 0x40u) return false;
        if (c & 0x80u) return true;
        if (i == 0) return false;
        return tk_certain_ctx(s->T, cls2[i - 1] & 15u, c & 15u, text.data(), i);
    };
    uint64_t n_walkback = 0;
    for (uint64_t t0 = 0; t0 < n; t0 += tile) {
        const uint64_t t1 = t0 + tile < n ? t0 + tile : n;
        auto scan_from = [&](uint64_t q) {
            for (;;) {
                uint64_t e = tk_piece_end(acc, q, patx);
                if (e <= q) e = tk_next_char(acc, q);
                if (e >= t1) break;                // the next piece start belongs to a later tile
                uint32_t ce = cls2[e];
                if (e >= t0) {
                    if (ce & 0x80u) break;
                    if (tk_certain_ctx(s->T, cls2[e - 1] & 15u, ce & 15u, text.data(), e)) break;  // a scanner of this tile starts there
                    starts[e] = 1;
                } else if (certain_at(e)) {
                    break;                            // cannot happen: q was the LAST certain start before the tile
                }
                q = e;
            }
        };
        // first char start of the tile
        uint64_t f = t0;
        while (f < t1 && (cls2[f] & 0x40u)) ++f;
        if (f < t1 && !certain_at(f)) {
            // last certain start before the tile: the left context first, then walk back
            uint64_t lo = t0 >= left ? t0 - left + 1 : 0;  // (the first context byte has no known predecessor)
            int64_t found = -1;
            for (int64_t j = (int64_t)t0 - 1; j >= (int64_t)lo; --j)
                if (certain_at((uint64_t)j)) {
                    found = j;
                    break;
                }
            if (found < 0) {
                ++n_walkback;
                int64_t j = (int64_t)lo - 1;
                while (j > 0 && !certain_at((uint64_t)j)) --j;
                found = j < 0 ? 0 : j;
            }
            scan_from((uint64_t)found);
        }
        for (uint64_t i = t0; i < t1; ++i)
            if (certain_at(i)) {
                starts[i] = 1;
                scan_from(i);
            }
    }
    return n_walkback;
}


// The "second stop" rule of the front kernel's phase D (tk_chunk.h, tk_chunk_second_stop), checked position by position against the
// sequential scanner: wherever the rule applies -- a certain start that is a one-byte char of a prefix class, the next byte a letter and an
// uncertain stop, the stop behind it certain -- the piece that starts there must end at that second stop.  Returns the violations;
// *applied gets the number of places where the rule applied.
uint64_t tks_second_stop_check(void* pv, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, uint64_t* applied) {
    Sim* s = (Sim*)pv;
    const int pat = s->T.pattern;
    std::vector<uint8_t> text(text_in, text_in + n);
    text.resize(n + 64, 0);
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) brk[doc_off[d] >> 5] |= 1u << (doc_off[d] & 31);
    std::vector<uint8_t> cls2(n + 80, TK_C_END | 0x80);
    uint8_t last = TK_C_OT;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = tk_class_byte(s->T, text.data(), i, n, brk.data(), nullptr, nullptr);
        if ((c & 15u) == TK_C_CONT) cls2[i] = (uint8_t)(last | 0x40);
        else {
            cls2[i] = (uint8_t)c;
            last = (uint8_t)(c & 15u);
        }
    }
    PropAcc acc{cls2.data(), text.data(), n};
    auto is_start = [&](uint64_t i) { return i >= n || !(cls2[i] & 0x40u); };
    auto certain_at = [&](uint64_t i) -> bool {
        if (i >= n) return true;
        uint32_t c = cls2[i];
        if (c & 0x40u) return false;
        if (c & 0x80u) return true;
        if (i == 0) return false;
        return tk_certain_ctx(s->T, cls2[i - 1] & 15u, c & 15u, text.data(), i);
    };
    auto near = [&](uint64_t i) { return (i >= 2 && text[i - 2] == '\'') || (i >= 3 && text[i - 3] == '\''); };
    auto stop_at = [&](uint64_t i) -> bool {
        if (i >= n) return true;
        if (!is_start(i)) return false;
        if (certain_at(i) || i == 0) return true;
        const uint32_t a = cls2[i - 1] & 15u, b = cls2[i] & 15u;
        return !(((tk_never_mask(pat, a) >> b) & 1u) && !near(i));
    };
    const uint32_t pc = s->T.pat.generic() ? 0u : tk_second_stop_prefix_classes(pat), lc = tk_second_stop_letter_classes(pat);
    uint64_t bad = 0, app = 0;
    for (uint64_t i = 0; i + 1 < n; ++i) {
        if (!is_start(i) || !is_start(i + 1) || !certain_at(i)) continue;
        if (!((pc >> (cls2[i] & 15u)) & 1u) || !((lc >> (cls2[i + 1] & 15u)) & 1u)) continue;
        if ((cls2[i + 1] & 0x80u) || certain_at(i + 1) || !stop_at(i + 1)) continue;
        uint64_t u2 = i + 2;
        while (u2 < n && !stop_at(u2)) ++u2;
        if (!certain_at(u2)) continue;
        ++app;
        uint64_t e = tk_piece_end(acc, i, s->T.pat);
        if (e <= i) e = tk_next_char(acc, i);
        if (e > n) e = n;
        if (e != u2) ++bad;
    }
    if (applied) *applied = app;
    return bad;
}

// Mirror of phases A-C of tk_k_front: the text is classified 16 bytes per "lane" with the functions of tk_chunk.h (table pass,
// decode pass, final masks, set algebra, certain starts) over every tile's 4096-byte window, and compared position by position
// with the per-byte reference (tk_class_byte + propagation, tk_certain_start).  ss / si (may be null): special-token bitmaps.
// Returns the number of mismatching positions; *first_bad gets the first one (text offset), *what a code (1 class, 2 cont,
// 3 hard, 4 certain).
uint64_t tks_chunk_check(void* pv, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, const uint32_t* ss,
                         const uint32_t* si, uint32_t tile, uint32_t left, uint32_t win, uint64_t* first_bad, uint32_t* what) {
    Sim* s = (Sim*)pv;
    const TkTables& T = s->T;
    std::vector<uint8_t> text(text_in, text_in + n);
    text.resize(n + 64, 0xA5);  // (garbage after the end, as a device buffer may hold: the kernel must mask it)
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) brk[doc_off[d] >> 5] |= 1u << (doc_off[d] & 31);
    if (ss)
        for (uint64_t i = 0; i < n; ++i)
            if (tk_bit(ss, i)) {
                brk[i >> 5] |= 1u << (i & 31);
                uint64_t e = i + 1;
                while (e < n && tk_bit(si, e)) ++e;
                if (e < n) brk[e >> 5] |= 1u << (e & 31);
            }
    // reference: class byte per position (continuation bytes carry their char's class | 0x40; 0x80 = hard)
    std::vector<uint8_t> ref(n + 80, TK_C_END | 0x80);
    uint8_t last = TK_C_OT;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = tk_class_byte(T, text.data(), i, n, brk.data(), ss, si);
        if ((c & 15u) == TK_C_CONT) ref[i] = (uint8_t)(last | 0x40);
        else {
            ref[i] = (uint8_t)c;
            last = (uint8_t)(c & 15u);
        }
    }
    const int pat = T.pattern;
    const uint16_t* s_cert = T.cert;
    (void)pat;
    const TkPat patx = T.pat;
    (void)patx;
    auto certain_ref = [&](uint64_t i, bool prev_known, bool ctx_known) -> bool {
        uint32_t c = ref[i];
        if (c & 0x40u) return false;
        if (c & 0x80u) return true;
        if (i == 0 || !prev_known) return false;
        // (the first three bytes of a window cannot see an apostrophe before it: only the table counts there)
        return ctx_known ? tk_certain_ctx(T, ref[i - 1] & 15u, c & 15u, text.data(), i) : tk_certain_rt(s_cert, ref[i - 1] & 15u, c & 15u);
    };
    uint64_t bad = 0;
    auto report = [&](uint64_t pos, uint32_t code) {
        if (!bad) {
            *first_bad = pos;
            *what = code;
        }
        ++bad;
    };
    const uint32_t nlanes = win / 16;
    for (uint64_t t0 = 0; t0 < n || t0 == 0; t0 += tile) {
        const int64_t base = (int64_t)t0 - left;
        std::vector<uint8_t> raw(win + 16, 0);
        std::vector<uint32_t> lastc(nlanes, 0);
        std::vector<TkChunkMasks> masks(nlanes);
        std::vector<TkSets> sets(nlanes);
        std::vector<uint32_t> valids(nlanes, 0);
        // pass 1: the window copy ("LDS")
        for (uint32_t t = 0; t < nlanes; ++t) {
            const int64_t gp = base + (int64_t)t * 16;
            for (int j = 0; j < 16; ++j) {
                const int64_t g = gp + j;
                raw[t * 16 + j] = (g >= 0 && (uint64_t)g < n) ? text[g] : 0;
            }
        }
        for (int j = 0; j < 16; ++j) {  // the 16 bytes behind the window, unmasked as in the kernel
            const int64_t g = base + win + j;
            raw[win + j] = (g >= 0 && (uint64_t)(base + win) < n && (uint64_t)g < n + 64) ? text[g] : 0;
        }
        for (uint32_t t = 0; t < nlanes; ++t) {
            const int64_t gp = base + (int64_t)t * 16;
            uint32_t w[4], valid = 0, past = 0;
            memcpy(w, &raw[t * 16], 16);
            for (int j = 0; j < 16; ++j) {
                const int64_t g = gp + j;
                if (g >= 0 && (uint64_t)g < n) valid |= 1u << j;
                if (g >= 0 && (uint64_t)g >= n) past |= 1u << j;
            }
            TkChunk ch;
            auto tab = [&](uint32_t b, uint32_t& x, uint32_t& y) {
                x = T.byte_tab[2 * b];
                y = T.byte_tab[2 * b + 1];
            };
            tk_chunk_table_pass(w, tab, ch);
            auto get4 = [&](int k) -> uint32_t {
                uint32_t v;
                memcpy(&v, &raw[(int64_t)t * 16 + k], 4);
                return v;
            };
            auto cls_of = [&](uint32_t cp) -> uint32_t { return tk_class_of_cp(T, cp > 0x10FFFFu ? 0xFFFFu : cp); };
            uint32_t prev = 0;
            if (t) memcpy(&prev, &raw[t * 16 - 4], 4);
            tk_chunk_decode(ch, prev, t > 0, get4, cls_of);
            uint32_t brk16 = 0, ss16 = 0, si16 = 0;
            for (int j = 0; j < 16; ++j) {
                const int64_t g = gp + j;
                if (g < 0 || (uint64_t)g >= n) continue;
                if (tk_bit(brk.data(), (uint64_t)g)) brk16 |= 1u << j;
                if (ss && tk_bit(ss, (uint64_t)g)) ss16 |= 1u << j;
                if (si && tk_bit(si, (uint64_t)g)) si16 |= 1u << j;
            }
            tk_chunk_finalize(ch, valid, past, brk16, ss16, si16, masks[t]);
            tk_sets_from_planes(masks[t].p[0], masks[t].p[1], masks[t].p[2], masks[t].p[3], sets[t]);
            lastc[t] = tk_class_from_planes(masks[t].p, 15);
            valids[t] = valid;
        }
        for (uint32_t t = 0; t < nlanes; ++t) {
            const int64_t gp = base + (int64_t)t * 16;
            const TkChunkMasks& m = masks[t];
            const uint32_t prevc = t ? lastc[t - 1] : 0u;
            uint32_t ap_before = t ? 0u : 7u;  // (the first chunk of a window cannot see what stands before it: as if apostrophes did)
            for (int b = 1; b <= 3; ++b)
                if (t > 0 && raw[(int64_t)t * 16 - b] == '\'') ap_before |= 1u << (3 - b);
            const uint32_t near16 = tk_chunk_near(sets[t].ap, ap_before);
            const uint32_t cert = T.pat.generic() ? tk_chunk_certain_rt(T.cert, sets[t], m.text, m.hard & m.text, prevc)
                                                   : tk_chunk_certain(pat, sets[t], m.text, m.hard & m.text, prevc, near16);
            // (the window may begin inside a char: its bytes there have no known class, and the first char start of the window no
            // known predecessor -- the kernel treats both as "unknown, never certain")
            int first = 0;
            if (t == 0)
                while (first < 16 && gp + first >= 0 && (uint64_t)(gp + first) < n && !((m.start >> first) & 1u)) ++first;
            uint32_t nev = 0;
            if (!T.pat.generic()) {
                nev = tk_chunk_never(pat, sets[t], prevc, near16);
            }
            for (int j = 0; j < 16; ++j) {
                const int64_t g = gp + j;
                if (g < 0 || (t == 0 && j < first)) continue;
                if (!T.pat.generic() && (uint64_t)g < n && !(t == 0 && j <= first)) {  // the set-algebra form against the table
                    bool want = false;
                    if ((m.start >> j) & 1u) {
                        int64_t q = g - 1;
                        while (q > 0 && (ref[q] & 0x40u)) --q;
                        want = q >= 0 && ((tk_never_mask(pat, ref[q] & 15u) >> (ref[g] & 15u)) & 1u);
                    }
                    const bool near = (g >= 2 && text_in[g - 2] == '\'') || (g >= 3 && text_in[g - 3] == '\'');
                    const bool got = (nev >> j) & 1u & ((m.start >> j) & 1u);
                    // claiming "never" where the table does not is an error; the other way round only without an apostrophe nearby
                    if ((got && (!want || near)) || (!got && want && !near && (t > 0 || j > 2))) report((uint64_t)g, 6);
                }
                const uint32_t cls = tk_class_from_planes(m.p, (uint32_t)j);
                const bool start = (m.start >> j) & 1u, hard = (m.hard >> j) & 1u;
                if ((uint64_t)g >= n) {  // past the end: END, a char start, hard
                    if (cls != TK_C_END || !start || !hard || ((cert >> j) & 1u)) report((uint64_t)g, 5);
                    continue;
                }
                const uint32_t r = ref[g];
                if (cls != (r & 15u)) report((uint64_t)g, 1);
                else if (start != !(r & 0x40u)) report((uint64_t)g, 2);
                else if (hard != (bool)(r & 0x80u)) report((uint64_t)g, 3);
                else if ((bool)((cert >> j) & 1u) != certain_ref((uint64_t)g, !(t == 0 && j == first), !(t == 0 && j < 3))) report((uint64_t)g, 4);
            }
        }
        if (n == 0) break;
    }
    return bad;
}

// Mirror of the per-piece work (whole-piece probe, lane merge) for pieces of <= 16 bytes; longer pieces use the
// same probes with a simple sequential merge over ids (the wave / tree kernels cannot run here).
int64_t tks_encode_piece(void* p, const uint8_t* piece, uint32_t len, uint32_t* out) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> text(piece, piece + len);
    text.resize(len + 64, 0);
    uint32_t r = tk_lookup_text_piece(s->T, text.data(), 0, len);
    if (r != TK_RANK_MAX) {
        out[0] = r;
        return 1;
    }
    if (len <= 128) {  // the production per-lane merge (tk_k_merge_llane), stride 1
        uint32_t s_id[128], s_rk[128];
        return tk_lane_merge<1>(s->T, text.data(), 0, len, s_id, s_rk, out);
    }
    std::vector<uint32_t> id(len), rk(len);
    for (uint32_t k = 0; k < len; ++k) {
        id[k] = s->T.byte_rank[text[k]];
        rk[k] = k + 1 < len ? s->T.pair2[((uint32_t)text[k] << 8) | text[k + 1]] : TK_RANK_MAX;
    }
    std::vector<uint32_t> ids(id), rks(rk);
    for (;;) {
        uint32_t best = TK_RANK_MAX;
        size_t bi = 0;
        for (size_t k = 0; k + 1 < ids.size(); ++k)
            if (rks[k] < best) {
                best = rks[k];
                bi = k;
            }
        if (best == TK_RANK_MAX) break;
        ids[bi] = best;
        ids.erase(ids.begin() + bi + 1);
        rks.erase(rks.begin() + bi + 1);
        rks[bi] = bi + 1 < ids.size() ? tk_probe_pair(s->T, ids[bi], ids[bi + 1]) : TK_RANK_MAX;
        if (bi > 0) rks[bi - 1] = tk_probe_pair(s->T, ids[bi - 1], ids[bi]);
    }
    for (size_t k = 0; k < ids.size(); ++k) out[k] = ids[k];
    return (int64_t)ids.size();
}

// ---- the generic pat_str engine (tk_regex.cpp, tk_regex_split.h): compile, then the two kernels' lanes one after the other
void* tks_rx_compile(const char* pat_str, char* err, uint64_t errcap) {
    TkRxCompiled* c = new TkRxCompiled();
    const std::string e = tk_rx_compile(pat_str, c);
    if (!e.empty()) {
        strncpy(err, e.c_str(), errcap - 1);
        err[errcap - 1] = 0;
        delete c;
        return nullptr;
    }
    return c;
}
void tks_rx_free(void* p) { delete (TkRxCompiled*)p; }
uint64_t tks_rx_size(void* p) { return ((TkRxCompiled*)p)->ins.size(); }
uint64_t tks_rx_steps() { return g_rx_steps; }
// the pattern's DFA (tk_regex_dfa.inc): states << 32 | classes, 0 when it has none (why: the reason, if `why` is given)
uint64_t tks_rx_dfa(void* p, char* why, uint64_t cap) {
    const TkRxCompiled* c = (const TkRxCompiled*)p;
    if (why && cap) {
        strncpy(why, c->dfa_why.c_str(), cap - 1);
        why[cap - 1] = 0;
    }
    return c->has_dfa() ? (uint64_t)c->dfa_nstates << 32 | c->dfa_ncls : 0;
}
// encode_mid's plan for one document (tk_mid_plan.h): the number of segments and cuts[0..k], or 0 and the reason.  Also 0 when the pattern's
// table does not make "letter, then space" a certain start (reason says so).
uint64_t tks_mid_plan(void* p, const uint8_t* text, uint64_t n, uint32_t* cuts /* [tks_mid_slots() + 1] */, char* why, uint64_t cap) {
    const Sim* s = (const Sim*)p;
    const char* reason = "";
    uint32_t k = 0;
    if (!s->H.rx.empty()) reason = "a pattern of the generic engine";
    else if (!tk_mid_cut_certain(s->H.cert)) reason = "letter -> space is not a certain start of this pattern";
    else k = tk_mid_plan(text, (uint32_t)n, cuts, &reason);
    if (why && cap) {
        strncpy(why, k ? "" : reason, cap - 1);
        why[cap - 1] = 0;
    }
    return k;
}
uint64_t tks_mid_slots() { return TK_SMALL_SLOTS; }
uint64_t tks_mid_segment_max() { return TK_MID_SEGMENT_MAX; }
uint64_t tks_mid_segments() { return TK_MID_SEGMENTS; }
}  // extern "C"
  // matcher steps so far (instructions + chars of repeats + backtracks)
// Piece starts of a packed batch: a byte per position (1 = start).  spec_at / spec_len: occurrences of allowed special tokens (sorted).
// speculate = 0: every document walked by the matcher alone; bits 0..1: 1 = speculative pass over 256-byte segments, 2 = over 1 KiB; bit 2: with
// the link pass; bit 3: documents resolved by groups of 64 lanes (the host form of the device's wavefront).
// bits 8..12: log2 of TkRxText::ahead (0: the default of the matcher's form).
// bit 4: the matcher is the pattern's DFA (it must have one: tks_rx_dfa) instead of the program; bit 5 (with bit 4): the speculative lanes in
// their one-loop form (tk_rx_speculate_lane_flat), compared bit for bit with the piece-by-piece form (error 0xFE if they differ), and
// once more over staged text (tk_rx_speculate_lane_codes: error 0xFD).
// stats[0] = matcher runs of the speculative (+ link) pass, [1] = of the resolving pass.  Returns 0, or error bits | position << 8.
static uint64_t g_rx_staged[2];  // lanes of the staged speculative pass that finished / that gave their segment to the one-loop lane
extern "C" void tks_rx_staged_stats(uint64_t* out, int reset) {
    out[0] = g_rx_staged[0];
    out[1] = g_rx_staged[1];
    if (reset) g_rx_staged[0] = g_rx_staged[1] = 0;
}
template <int DFA>  // 0: the program; 1: the pattern's table; 2: the table of a pattern that looks behind
static uint64_t rx_split_impl(void* p, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, const uint64_t* spec_at,
                              const uint64_t* spec_len, uint64_t n_spec, int speculate, uint8_t* starts, uint64_t* stats) {
    const uint32_t seg_shift = (speculate & 3) == 2 ? TK_RX_SEG_SHIFT_LARGE : TK_RX_SEG_SHIFT_SMALL;
    const bool with_links = (speculate & 3) && (speculate & 4), by_group = (speculate & 8) != 0;
    const TkRxCompiled* c = (const TkRxCompiled*)p;
    const TkRxProg P = c->view();
    std::vector<uint8_t> text(text_in, text_in + n);
    text.resize(n + 64, 0);
    const uint64_t nw = (n + 31) / 32 + 2;
    std::vector<uint32_t> brk(nw, 0), ss(nw, 0), si(nw, 0), spec(nw, 0), sgap(nw, 0), gst(nw, 0), ggap(nw, 0), lnk(nw, 0), lgap(nw, 0);
    auto setb = [](std::vector<uint32_t>& v, uint64_t q) { v[q >> 5] |= 1u << (q & 31); };
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) setb(brk, doc_off[d]);
    for (uint64_t k = 0; k < n_spec; ++k) {
        setb(ss, spec_at[k]);
        setb(brk, spec_at[k]);
        for (uint64_t j = spec_at[k] + 1; j < spec_at[k] + spec_len[k]; ++j) setb(si, j);
        if (spec_at[k] + spec_len[k] < n) setb(brk, spec_at[k] + spec_len[k]);
    }
    TkRxText t{text.data(), (uint32_t)n, brk.data(), n_spec ? ss.data() : nullptr, n_spec ? si.data() : nullptr, 0xFFFFFFFFu, false};
    if (speculate >> 8) t.ahead = 1u << ((speculate >> 8) & 31);  // (bits 8..12: log2 of the bytes a speculative match may look beyond its segment; 0: the program's default)
    else if (DFA) t.ahead = TK_RX_AHEAD_DFA;
    const uint32_t nseg = (uint32_t)((n + (1u << seg_shift) - 1) >> seg_shift);
    std::vector<uint32_t> xexit(nseg + 1, TK_RX_UNKNOWN), lmerge(nseg + 1, TK_RX_NOLINK), lexit(nseg + 1, TK_RX_UNKNOWN);
    g_rx_matches = 0;
    if ((speculate & 3) && (speculate & 32)) {  // the one-loop form of the DFA's speculative lane: the same bitmaps and exits, bit for bit
        if constexpr (DFA) {
            for (uint32_t k = 0; k < nseg; ++k) tk_rx_speculate_lane_flat<DFA == TK_RX_M_DFA_PREV>(P, t, k, seg_shift, spec.data(), sgap.data(), xexit.data());
            std::vector<uint32_t> spec2(nw, 0), sgap2(nw, 0), xexit2(nseg + 1, TK_RX_UNKNOWN);
            for (uint32_t k = 0; k < nseg; ++k) tk_rx_speculate_lane<DFA>(P, t, k, seg_shift, spec2.data(), sgap2.data(), xexit2.data());
            if (spec2 != spec || sgap2 != sgap || xexit2 != xexit) return 0xFEu;
            // ... and the lanes over STAGED text (tk_k_rx_speculate_staged: a workgroup's 256 segments as codes, the one-loop lane wherever a lane gives up)
            if constexpr (DFA == TK_RX_M_DFA) {
                if (seg_shift == TK_RX_SEG_SHIFT_SMALL && P.dfa_ncls <= TK_RX_CODE_MAX_CLS + 1u && c->dfa_trans.size() < 32768u) {
                    std::vector<uint32_t> spec3(nw, 0), sgap3(nw, 0), xexit3(nseg + 1, TK_RX_UNKNOWN);
                    std::vector<uint8_t> codes(TK_RX_STAGE_BYTES);
                    std::vector<uint16_t> trans_pm(c->dfa_trans.size());
                    for (size_t i = 0; i < trans_pm.size(); ++i) trans_pm[i] = tk_rx_trans_premultiplied(c->dfa_trans[i], P.dfa_ncls);
                    for (uint32_t s0 = 0; s0 < nseg; s0 += TK_RX_STAGE_SEGS) {
                        const uint32_t r0 = s0 << TK_RX_SEG_SHIFT_SMALL;
                        for (uint32_t b = 0; b < TK_RX_STAGE_BYTES; b += 16) {
                            uint32_t o[4] = {0, 0, 0, 0};
                            if ((uint64_t)r0 + b < n) tk_rx_codes16(P, t, r0 + b, o);
                            tk_rx_codes_store(codes.data(), b, o);
                        }
                        const TkRxCodes C{codes.data(), r0, TK_RX_STAGE_BYTES};
                        for (uint32_t k = s0; k < s0 + TK_RX_STAGE_SEGS && k < nseg; ++k) {
                            uint32_t sb[4], gb[4], x;
                            if (tk_rx_speculate_lane_codes(trans_pm.data(), P.dfa_ncls, C, (uint32_t)n, t.ahead, k, sb, gb, &x)) {
                                for (uint32_t i = 0; i < 4; ++i) {
                                    if (sb[i]) spec3[4 * (size_t)k + i] |= sb[i];
                                    if (gb[i]) sgap3[4 * (size_t)k + i] |= gb[i];
                                }
                                xexit3[k] = x;
                                ++g_rx_staged[0];
                            } else {
                                tk_rx_speculate_lane_flat<false>(P, t, k, seg_shift, spec3.data(), sgap3.data(), xexit3.data());
                                ++g_rx_staged[1];
                            }
                        }
                    }
                    if (spec3 != spec || sgap3 != sgap || xexit3 != xexit) return 0xFDu;
                }
            }
        } else {
            return 0xFFu;
        }
    } else if (speculate & 3)
        for (uint32_t k = 0; k < nseg; ++k) tk_rx_speculate_lane<DFA>(P, t, k, seg_shift, spec.data(), sgap.data(), xexit.data());
    if (with_links)
        for (uint32_t k = 0; k < nseg; ++k) tk_rx_link_lane<DFA>(P, t, k, seg_shift, spec.data(), xexit.data(), lnk.data(), lgap.data(), lmerge.data(), lexit.data());
    stats[0] = g_rx_matches;
    g_rx_matches = 0;
    const TkRxMaps M{(speculate & 3) ? spec.data() : nullptr, sgap.data(), xexit.data(), with_links ? lnk.data() : nullptr, lgap.data(), lmerge.data(), lexit.data(), seg_shift};
    uint64_t rc = 0;
    for (uint64_t d = 0; d < n_docs && !rc; ++d) {
        uint32_t err_pos = 0;
        auto orb = [&](uint32_t w, uint32_t bits, uint32_t gaps) {
            gst[w] |= bits;
            ggap[w] |= gaps;
        };
        const uint32_t e = by_group ? tk_rx_resolve_group_host<DFA>(P, t, M, (uint32_t)doc_off[d], (uint32_t)doc_off[d + 1], orb, &err_pos)
                                    : tk_rx_resolve_lane<DFA>(P, t, M, (uint32_t)doc_off[d], (uint32_t)doc_off[d + 1], orb, &err_pos);
        if (e) rc = e | ((uint64_t)err_pos << 8);
    }
    stats[1] = g_rx_matches;
    for (uint64_t i = 0; i < n; ++i) starts[i] = ((gst[i >> 5] >> (i & 31)) & 1u) | (((ggap[i >> 5] >> (i & 31)) & 1u) << 1);  // bit 1: a gap char
    return rc;
}

extern "C" uint64_t tks_rx_split(void* p, const uint8_t* text_in, uint64_t n, const uint64_t* doc_off, uint64_t n_docs, const uint64_t* spec_at,
                                 const uint64_t* spec_len, uint64_t n_spec, int speculate, uint8_t* starts, uint64_t* stats) {
    if (speculate & 16) {
        if (!((const TkRxCompiled*)p)->has_dfa()) return 0xFFu;
        if (((const TkRxCompiled*)p)->dfa_flags & 1u) return rx_split_impl<TK_RX_M_DFA_PREV>(p, text_in, n, doc_off, n_docs, spec_at, spec_len, n_spec, speculate, starts, stats);
        return rx_split_impl<TK_RX_M_DFA>(p, text_in, n, doc_off, n_docs, spec_at, spec_len, n_spec, speculate, starts, stats);
    }
    return rx_split_impl<TK_RX_M_PROGRAM>(p, text_in, n, doc_off, n_docs, spec_at, spec_len, n_spec, speculate, starts, stats);
}
