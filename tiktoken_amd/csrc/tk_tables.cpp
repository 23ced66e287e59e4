#include "tk_tables.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <numeric>
#include <thread>

int tk_pattern_id(const char* pat_str) {
    TkPat p;
    TkRxCompiled rx;
    if (!tk_compile_pattern(pat_str, &p, nullptr, &rx).empty()) return -1;
    return rx.empty() ? p.fam() : 3;
}

// r50k family, generic kernels, no contractions: with every char classed as a letter its only alternative that ever matches is \p{L}+,
// which runs from one hard start to the next
TkPat tk_nosplit_pat() { return TkPat{(uint32_t)TK_PAT_R50K | 32u, 0u, {0u, 0u}}; }

std::string tk_compile_pattern(const char* pat_str, TkPat* pat, uint16_t* cert, TkRxCompiled* rx) {
    // (TIKTOKEN_AMD_DEBUG bit 0x100000 = 1048576: the generic engine for every pattern -- how the tests compare it with the hand-written scanners)
    const char* dbg = getenv("TIKTOKEN_AMD_DEBUG");
    const bool force_generic = dbg && (atoi(dbg) & 0x100000);
    const std::string perr = force_generic ? std::string("generic engine forced by TIKTOKEN_AMD_DEBUG") : tk_parse_pattern(pat_str, pat, cert);
    if (perr.empty()) return "";
    const std::string rerr = tk_rx_compile(pat_str, rx);
    if (!rerr.empty()) {
        rx->ins.clear();
        return "pat_str is not supported: " + rerr + " [and it is not one of the hand-written scanner families: " + perr + "]";
    }
    *pat = tk_nosplit_pat();
    if (cert) memset(cert, 0, 16 * sizeof(uint16_t));  // no certain starts but the hard ones
    return "";
}

uint64_t tk_key_of_bytes(const uint8_t* p, uint32_t len) {
    if (len <= 8) {
        uint64_t k = 0;
        memcpy(&k, p, len);
        return k;
    }
    if (len > TK_KEY_SAMPLED) {
        uint64_t h = TK_HASH_SEED ^ ((uint64_t)len << 32);
        for (uint32_t o : {0u, 8u, len - 16u, len - 8u}) {
            uint64_t w;
            memcpy(&w, p + o, 8);
            h = tk_hash_step(h, w);
        }
        return h == TK_EMPTY_KEY ? 0 : h;
    }
    uint64_t h = TK_HASH_SEED;
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = tk_hash_step(h, w);
    }
    if (i < len) {
        uint64_t w = 0;
        memcpy(&w, p + i, len - i);
        h = tk_hash_step(h, w);
    }
    if (h == TK_EMPTY_KEY) h = 0;  // reserve the empty marker
    return h;
}

uint32_t TkHostTables::lookup_piece(const uint8_t* p, uint32_t len) const {
    if (len == 0) return TK_RANK_MAX;
    if (len <= 8) {
        uint64_t key = 0;
        memcpy(&key, p, len);
        if (len <= 4 && !short_tab.empty()) {
            uint32_t i = tk_short_slot((uint32_t)key, short_shift);
            for (;;) {
                const TkShortSlot& s = short_tab[i];
                if (s.val == TK_SHORT_EMPTY) return TK_RANK_MAX;
                if (s.key == (uint32_t)key && (s.val >> 30) == len - 1) return s.val & TK_SHORT_MAX_RANK;
                i = (i + 1) & short_mask;
            }
        }
        if (mid_tab.empty()) return TK_RANK_MAX;
        uint32_t i = tk_mid_slot(key, mid_shift);
        for (;;) {
            const TkPieceSlot& s = mid_tab[i];
            if (s.len == 0) return TK_RANK_MAX;
            if (s.key == key && s.len == len) return s.rank;
            i = (i + 1) & mid_mask;
        }
    }
    if (piece.empty()) return TK_RANK_MAX;
    uint64_t key = tk_key_of_bytes(p, len);
    uint64_t i = tk_piece_slot_hash(key, len) & piece_mask;
    for (;;) {
        const TkPieceSlot& s = piece[i];
        if (s.key == TK_EMPTY_KEY && s.len == 0) return TK_RANK_MAX;
        if (s.key == key && s.len == len) {
            if (memcmp(tok_bytes.data() + piece_off[i], p, len) == 0) return s.rank;
        }
        i = (i + 1) & piece_mask;
    }
}

uint32_t TkHostTables::lookup_pair(uint32_t a, uint32_t b) const {
    if (!pair8.empty()) {
        const uint64_t key = ((uint64_t)a << TK_PAIR8_ID_BITS) | b;
        uint64_t bk = tk_pair_slot_hash(key) & pair_mask;
        for (;;) {
            for (int j = 0; j < 4; ++j) {
                uint64_t s = pair8[bk * 4 + j];
                if ((s >> 22) == key) return (uint32_t)(s & 0x3FFFFFu);
                if (s == TK_EMPTY_KEY) return TK_RANK_MAX;
            }
            bk = (bk + 1) & pair_mask;
        }
    }
    uint64_t key = ((uint64_t)a << 32) | b;
    uint64_t i = tk_pair_slot_hash(key) & pair_mask;
    for (;;) {
        const TkPairSlot& s = pair[i];
        if (s.key == TK_EMPTY_KEY) return TK_RANK_MAX;
        if (s.key == key) return s.rank;
        i = (i + 1) & pair_mask;
    }
}

std::string tk_build_tables(const uint8_t* ranks_blob, const uint64_t* ranks_off, const uint32_t* ranks_ids,
                            uint64_t n_ranks, const uint8_t* spec_blob, const uint64_t* spec_off,
                            const uint32_t* spec_ids, uint64_t n_spec, const char* pat_str, TkHostTables* out) {
    TkHostTables& T = *out;
    const bool timing = getenv("TIKTOKEN_AMD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "tk_build_tables: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    {
        const std::string perr = tk_compile_pattern(pat_str, &T.pat, T.cert, &T.rx);
        if (!perr.empty()) return perr;
        T.pattern = T.pat.fam();
    }
    lap("pattern");
    if (n_ranks == 0) return "mergeable_ranks is empty";
    if (ranks_off[n_ranks] >= 0xFFFFFFFFull) return "vocabulary byte blob too large";
    T.n_ranks = n_ranks;
    T.tok_bytes.assign(ranks_blob, ranks_blob + ranks_off[n_ranks]);
    T.tok_bytes.resize(T.tok_bytes.size() + 16, 0);  // device verification reads whole 8-byte words

    // piece tables (bytes -> rank), split by length
    uint32_t max_rank = 0;
    uint64_t n_short = 0, n_mid = 0, n_long = 0, n_xl = 0;
    for (uint64_t k = 0; k < n_ranks; ++k) {
        max_rank = std::max(max_rank, ranks_ids[k] == TK_RANK_MAX ? 0u : ranks_ids[k]);
        const uint64_t len = ranks_off[k + 1] - ranks_off[k];
        (len <= 4 ? n_short : (len <= 8 ? n_mid : n_long)) += 1;
        if (len >= TK_XL_MIN && len <= TK_XL_MAX) ++n_xl;
    }
    for (uint64_t k = 0; k < n_spec; ++k) max_rank = std::max(max_rank, spec_ids[k]);
    if (max_rank >= 0x7FFFFFFFu)
        return "token ids of 2^31 - 1 and above are not supported (the per-piece result word keeps its top bit for pieces that are not a "
               "single token, and one value for text that yields no token)";
    const bool use_short = max_rank <= TK_SHORT_MAX_RANK;
    if (!use_short) {
        n_mid += n_short;
        n_short = 0;
    }
    auto pow2_for = [](uint64_t items, uint32_t* mask, uint32_t* shift) {  // load factor <= 0.5
        uint32_t bits = 6;
        while ((1ull << bits) < 2 * items + 2) ++bits;
        *mask = (uint32_t)((1ull << bits) - 1);
        *shift = 32 - bits;
        return (uint64_t)1 << bits;
    };
    if (use_short) T.short_tab.assign(pow2_for(n_short, &T.short_mask, &T.short_shift), TkShortSlot{0xFFFFFFFFu, TK_SHORT_EMPTY});
    T.mid_tab.assign(pow2_for(n_mid, &T.mid_mask, &T.mid_shift), TkPieceSlot{TK_EMPTY_KEY, TK_RANK_MAX, 0});
    uint64_t cap = 64;
    while (cap < 2 * n_long + 2) cap <<= 1;
    T.piece_mask = cap - 1;
    T.piece.assign(cap, TkPieceSlot{TK_EMPTY_KEY, TK_RANK_MAX, 0});
    T.piece_off.assign(cap, 0);
    {  // (load factor <= 0.4: three in five probes of this table are for pieces that are not tokens, and such a probe ends at a free slot)
        uint64_t xcap = 64;
        while (2 * xcap < 5 * n_xl + 2) xcap <<= 1;
        T.xl_mask = (uint32_t)(xcap - 1);
        T.xl.assign(xcap, TkXlSlot{~0ull, ~0ull, ~0ull, TK_RANK_MAX, 0u});
        T.xfilter.assign(TK_XFILTER_BITS / 32, 0u);
    }
    for (int b = 0; b < 256; ++b) T.byte_rank[b] = TK_RANK_MAX;
    T.pair2.assign(65536, TK_RANK_MAX);
    {
        uint32_t mr = 0;
        for (uint64_t k = 0; k < n_ranks; ++k) mr = std::max(mr, ranks_ids[k] == TK_RANK_MAX ? 0u : ranks_ids[k]);
        T.max_rank = mr;
        T.dec_is_dense = (uint64_t)mr < 4 * n_ranks + 65536;
        if (T.dec_is_dense) T.dec_dense.assign((size_t)mr + 1, std::make_pair(0u, 0u));
        else T.dec_sparse.reserve(n_ranks * 2);
    }
    uint64_t pr_short = 0, pr_mid = 0, pr_long = 0, pr_xl = 0;
    for (uint64_t k = 0; k < n_ranks; ++k) {
        uint64_t o = ranks_off[k], len64 = ranks_off[k + 1] - o;
        if (len64 == 0) return "mergeable_ranks contains an empty key";
        uint32_t len = (uint32_t)len64, rank = ranks_ids[k];
        if (rank == TK_RANK_MAX) return "rank 0xFFFFFFFF is reserved (Rank::MAX sentinel, src/lib.rs:53)";
        const uint8_t* p = ranks_blob + o;
        if (T.lookup_piece(p, len) != TK_RANK_MAX) return "duplicate key in mergeable_ranks";
        bool fresh;
        if (T.dec_is_dense) {
            fresh = T.dec_dense[rank].second == 0;
            T.dec_dense[rank] = std::make_pair((uint32_t)o, len);
        } else {
            fresh = T.dec_sparse.emplace(rank, std::make_pair((uint32_t)o, len)).second;
        }
        if (!fresh)
            return "Encoder and decoder must be of equal length. Maybe you had duplicate token indices in your "
                   "encoder?";  // src/lib.rs:636-641
        if (len <= 8) {
            uint64_t key = 0;
            memcpy(&key, p, len);
            if (len <= 4 && use_short) {
                uint32_t i = tk_short_slot((uint32_t)key, T.short_shift);
                ++pr_short;
                while (T.short_tab[i].val != TK_SHORT_EMPTY) {
                    i = (i + 1) & T.short_mask;
                    ++pr_short;
                }
                T.short_tab[i] = TkShortSlot{(uint32_t)key, rank | ((len - 1) << 30)};
            } else {
                uint32_t i = tk_mid_slot(key, T.mid_shift);
                ++pr_mid;
                while (T.mid_tab[i].len != 0) {
                    i = (i + 1) & T.mid_mask;
                    ++pr_mid;
                }
                T.mid_tab[i] = TkPieceSlot{key, rank, len};
            }
        } else {
            uint64_t key = tk_key_of_bytes(p, len);
            uint64_t i = tk_piece_slot_hash(key, len) & T.piece_mask;
            ++pr_long;
            while (!(T.piece[i].key == TK_EMPTY_KEY && T.piece[i].len == 0)) {
                i = (i + 1) & T.piece_mask;
                ++pr_long;
            }
            T.piece[i] = TkPieceSlot{key, rank, len};
            T.piece_off[i] = (uint32_t)o;
            uint64_t w0, w1, w2;
            tk_ident([&](uint32_t at) {  // (eight bytes at offset `at`; what lies behind the token is masked away by tk_ident)
                uint64_t w = 0;
                memcpy(&w, p + at, at < len ? std::min<uint32_t>(8u, len - at) : 0u);
                return w;
            }, len, 0u, w0, w1, w2);
            if (len > TK_XL_MAX) {
                uint64_t hh = tk_ident_hash(w0, w1, w2, false);
                if (hh == TK_EMPTY_KEY) hh = 0;  // (as the front kernel stores it)
                const uint32_t bit = tk_xfilter_bit(hh);
                T.xfilter[bit >> 5] |= 1u << (bit & 31u);
            }
            if (len >= TK_XL_MIN && len <= TK_XL_MAX) {
                uint32_t j = (uint32_t)tk_ident_hash(w0, w1, w2, true) & T.xl_mask;
                ++pr_xl;
                while (T.xl[j].rank != TK_RANK_MAX) {
                    j = (j + 1) & T.xl_mask;
                    ++pr_xl;
                }
                T.xl[j] = TkXlSlot{w0, w1, w2, rank, 0u};
            }
        }
        if (len == 1) T.byte_rank[p[0]] = rank;
        if (len == 2) T.pair2[((uint32_t)p[0] << 8) | p[1]] = rank;
        if (len > T.max_token_len) T.max_token_len = len;
    }
    T.probes_short = n_short ? (double)pr_short / (double)n_short : 0;
    T.probes_mid = n_mid ? (double)pr_mid / (double)n_mid : 0;
    T.probes_long = n_long ? (double)pr_long / (double)n_long : 0;
    T.probes_xl = n_xl ? (double)pr_xl / (double)n_xl : 0;
    for (int b = 0; b < 256; ++b)
        if (T.byte_rank[b] == TK_RANK_MAX)
            return "every single byte must be a key of mergeable_ranks (byte_pair_encode indexes ranks[piece] for "
                   "1-byte pieces, src/lib.rs:201-203)";

    lap("piece tables + decoder");
    // pair table: all splits of all tokens into two vocabulary tokens (read-only lookups: split over host threads)
    std::vector<TkPairSlot> entries;
    {
        unsigned nth = std::thread::hardware_concurrency();
        nth = nth == 0 ? 1 : (nth > 32 ? 32 : nth);
        if (const char* e = getenv("TIKTOKEN_AMD_BUILD_THREADS")) nth = (unsigned)std::max(1, atoi(e));
        if (n_ranks < 4096) nth = 1;
        // blocks of 256 tokens, dealt round-robin (the file is in rank order: its tail holds the long tokens -- most of the work); the
        // entries are put together in block order, so the table's layout does not depend on the number of threads
        const uint64_t nblocks = (n_ranks + 255) / 256;
        std::vector<std::vector<TkPairSlot>> part(nblocks);
        auto work = [&](unsigned t) {
            for (uint64_t b = t; b < nblocks; b += nth) {
                std::vector<TkPairSlot>& e = part[b];
                for (uint64_t k = b * 256; k < b * 256 + 256 && k < n_ranks; ++k) {
                    uint64_t o = ranks_off[k];
                    uint32_t len = (uint32_t)(ranks_off[k + 1] - o), rank = ranks_ids[k];
                    const uint8_t* p = ranks_blob + o;
                    for (uint32_t s = 1; s < len; ++s) {
                        uint32_t a = T.lookup_piece(p, s);
                        if (a == TK_RANK_MAX) continue;
                        uint32_t b2 = T.lookup_piece(p + s, len - s);
                        if (b2 == TK_RANK_MAX) continue;
                        e.push_back(TkPairSlot{((uint64_t)a << 32) | b2, rank, 0});
                    }
                }
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nth; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        size_t tot = 0;
        for (auto& e : part) tot += e.size();
        entries.reserve(tot);
        for (auto& e : part) entries.insert(entries.end(), e.begin(), e.end());
    }
    T.n_pairs = entries.size();
    lap("pair entries");

    uint32_t max_id = 0;
    for (uint64_t k = 0; k < n_ranks; ++k) max_id = std::max(max_id, ranks_ids[k]);
    if (max_id <= TK_PAIR8_MAX_ID) {
        // packed: 4-slot buckets, load factor <= 0.5; each key goes to the first bucket of its probe
        // sequence with a free slot, slots of a bucket fill in order
        uint64_t nb = 16;
        while ((double)nb * 4 * 0.5 < (double)entries.size() + 1) nb <<= 1;
        T.pair_mask = nb - 1;
        T.pair8.assign(nb * 4, TK_EMPTY_KEY);
        for (const TkPairSlot& e : entries) {
            const uint64_t key = ((e.key >> 32) << TK_PAIR8_ID_BITS) | (e.key & 0xFFFFFFFFull);
            uint64_t bk = tk_pair_slot_hash(key) & T.pair_mask;
            for (;;) {
                int j = 0;
                while (j < 4 && T.pair8[bk * 4 + j] != TK_EMPTY_KEY) ++j;
                if (j < 4) {
                    T.pair8[bk * 4 + j] = (key << 22) | e.rank;
                    break;
                }
                bk = (bk + 1) & T.pair_mask;
            }
        }
    } else {
        cap = 64;
        while ((double)cap * 0.4 < (double)entries.size() + 1) cap <<= 1;
        T.pair_mask = cap - 1;
        T.pair.assign(cap, TkPairSlot{TK_EMPTY_KEY, TK_RANK_MAX, 0});
        for (const TkPairSlot& e : entries) {
            uint64_t i = tk_pair_slot_hash(e.key) & T.pair_mask;
            while (T.pair[i].key != TK_EMPTY_KEY) i = (i + 1) & T.pair_mask;
            T.pair[i] = e;
        }
    }

    lap("pair table");
    // special tokens, sorted by bytes (deterministic order; the reference's alternation order is
    // hash-map order, src/lib.rs:625-631)
    std::vector<uint64_t> order(n_spec);
    std::iota(order.begin(), order.end(), 0);
    auto sbytes = [&](uint64_t k) { return std::string((const char*)spec_blob + spec_off[k], spec_off[k + 1] - spec_off[k]); };
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return sbytes(a) < sbytes(b); });
    memset(T.spec_first, 0, sizeof T.spec_first);
    T.spec_off.push_back(0);
    for (uint64_t idx : order) {
        std::string s = sbytes(idx);
        if (s.empty()) return "empty special token";
        // Two special strings may share an id (o200k_harmony: <|endofprompt|> and <|reserved_200018|> are both 200018,
        // openai_public.py:85-94); the reference's decoder map keeps one of them (HashMap collect, lib.rs:643-646), here the
        // first in byte order.
        T.spec_decoder.emplace(spec_ids[idx], std::make_pair((uint32_t)T.spec_bytes.size(), (uint32_t)s.size()));
        T.spec_bytes.insert(T.spec_bytes.end(), s.begin(), s.end());
        T.spec_off.push_back((uint32_t)T.spec_bytes.size());
        T.spec_id.push_back(spec_ids[idx]);
        unsigned char f = (unsigned char)s[0];
        T.spec_first[f >> 5] |= 1u << (f & 31);
    }
    for (size_t i = 1; i < order.size(); ++i)
        if (sbytes(order[i]) == sbytes(order[i - 1])) return "duplicate special token string";
    T.spec_bytes.resize(T.spec_bytes.size() + 16, 0);

    lap("special tokens");
    return "";
}

// sorted token bytes (token_byte_values, src/lib.rs:648-650, src/py.rs:178-183)
const std::vector<uint32_t>& TkHostTables::sorted_ranks() const {
    std::call_once(sorted_once_, [this] {
        struct E {
            uint32_t rank, off, len;
        };
        std::vector<E> v;
        v.reserve(n_ranks);
        for_each_token([&](uint32_t r, uint32_t o, uint32_t l) { v.push_back(E{r, o, l}); });
        const uint8_t* b = tok_bytes.data();
        std::sort(v.begin(), v.end(), [b](const E& x, const E& y) {
            const int c = memcmp(b + x.off, b + y.off, x.len < y.len ? x.len : y.len);
            return c < 0 || (c == 0 && x.len < y.len);
        });
        sorted_ranks_.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) sorted_ranks_[i] = v[i].rank;
    });
    return sorted_ranks_;
}


// ------------------------------------------------------------------------------------------
// `.tiktoken` wire format (reference tiktoken/load.py:159-171: one `base64(token) SP rank` per line), parsed natively:
// the stock o200k file is 200 k lines, which the reference parses in a Python loop.
// ------------------------------------------------------------------------------------------
std::string tk_parse_tiktoken(const uint8_t* text, uint64_t len, std::vector<uint8_t>* blob, std::vector<uint64_t>* off,
                              std::vector<uint32_t>* ids) {
    // As lenient as the reference's three Python calls per line are (tiktoken/load.py:162-171: `contents.splitlines()`, `line.split()`,
    // `base64.b64decode(token)`, `int(rank)`), so that a file the reference loads loads here as well:
    //   * lines end at \n, \r or \r\n; empty lines are skipped;
    //   * a line is exactly two fields between runs of ASCII white space (blank, \t, \n, \r, \v, \f), leading and trailing runs allowed;
    //   * base64 the way binascii.a2b_base64 reads it when it does not validate: bytes outside the alphabet are skipped (a byte-order
    //     mark, a '!'), a '=' ends the data once it completes a quad's padding and is skipped before that, what follows is ignored,
    //     and data that stops one or two sextets into a quad without padding is an error;
    //   * the rank as int() reads it: an optional sign, decimal digits, single underscores between digits.  A negative rank or one
    //     beyond 32 bits is an error HERE (the reference gets as far as CoreBPE's constructor with it: OverflowError, src/py.rs:20).
    static const std::array<int8_t, 256> dec = [] {
        std::array<int8_t, 256> d;
        d.fill(-1);
        const char* al = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) d[(unsigned char)al[i]] = (int8_t)i;
        return d;
    }();
    auto is_ws = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); };
    blob->clear();
    off->assign(1, 0);
    ids->clear();
    blob->reserve(len / 2);
    uint64_t line_no = 0, i = 0;
    auto bad = [&](const char* what) { return std::string("Error parsing line ") + std::to_string(line_no) + " of the .tiktoken data: " + what; };
    while (i < len) {
        ++line_no;
        uint64_t e = i;
        while (e < len && text[e] != '\n' && text[e] != '\r') ++e;
        const uint64_t a0 = i, b0 = e;
        i = e + 1;
        if (e < len && text[e] == '\r' && i < len && text[i] == '\n') ++i;
        if (a0 == b0) continue;  // empty line
        // the two fields
        uint64_t a = a0;
        while (a < b0 && is_ws(text[a])) ++a;
        uint64_t sp = a;
        while (sp < b0 && !is_ws(text[sp])) ++sp;
        uint64_t ra = sp;
        while (ra < b0 && is_ws(text[ra])) ++ra;
        uint64_t rb = ra;
        while (rb < b0 && !is_ws(text[rb])) ++rb;
        uint64_t rest = rb;
        while (rest < b0 && is_ws(text[rest])) ++rest;
        if (a == sp || ra == rb || rest != b0) return bad("expected `base64 SP rank`");
        // the token
        const size_t blob_before = blob->size();
        uint32_t acc = 0, quad = 0, pads = 0;
        bool done = false;
        for (uint64_t k = a; k < sp && !done; ++k) {
            const uint8_t ch = text[k];
            if (ch == '=') {
                if (quad >= 2 && quad + ++pads >= 4) done = true;  // (the quad's data bytes are out already)
                continue;
            }
            const int v = dec[ch];
            if (v < 0) continue;
            pads = 0;
            switch (quad) {
                case 0: acc = (uint32_t)v; quad = 1; break;
                case 1: blob->push_back((uint8_t)((acc << 2) | ((uint32_t)v >> 4))); acc = (uint32_t)v & 15u; quad = 2; break;
                case 2: blob->push_back((uint8_t)((acc << 4) | ((uint32_t)v >> 2))); acc = (uint32_t)v & 3u; quad = 3; break;
                default: blob->push_back((uint8_t)((acc << 6) | (uint32_t)v)); acc = 0; quad = 0; break;
            }
        }
        if (!done && quad != 0) {
            blob->resize(blob_before);
            return bad(quad == 1 ? "base64: one character more than a multiple of four" : "base64: incorrect padding");
        }
        // the rank
        uint64_t k = ra;
        bool neg = false;
        if (text[k] == '+' || text[k] == '-') neg = text[k++] == '-';
        if (k == rb || text[k] < '0' || text[k] > '9') return bad("rank is not a decimal number");
        uint64_t r = 0;
        bool wide = false;
        for (; k < rb; ++k) {
            if (text[k] == '_') {  // (between two digits only)
                if (k + 1 == rb || text[k + 1] < '0' || text[k + 1] > '9') return bad("rank is not a decimal number");
                continue;
            }
            if (text[k] < '0' || text[k] > '9') return bad("rank is not a decimal number");
            r = r * 10 + (uint64_t)(text[k] - '0');
            if (r > 0xFFFFFFFFull) wide = true, r = 0x100000000ull;  // (keep reading: "not a number" comes first if it is not one)
        }
        if (wide) return bad("rank does not fit 32 bits");
        if (neg && r != 0) return bad("rank is negative");
        off->push_back(blob->size());
        ids->push_back((uint32_t)r);
    }
    return "";
}
