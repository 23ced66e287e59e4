// Host-side construction of the device tables (the analogue of CoreBPE::new_internal,
// reference src/lib.rs:618-663: build encoder/decoder maps, check for duplicate ranks).
#pragma once
#include <stdint.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "tk_common.h"
#include "tk_regex_host.h"

struct TkHostTables {
    int pattern = -1;  // family of the split pattern
    TkPat pat{};       // the pattern (tk_pattern.cpp)
    uint16_t cert[16] = {0};  // its certain piece starts
    TkRxCompiled rx;          // a pat_str outside the scanner families: its program for the generic engine (tk_regex.cpp); pat is then
                              // the "no split" member of the r50k family, run over a class table in which every char is a letter
    std::vector<uint8_t> tok_bytes;
    std::vector<TkShortSlot> short_tab;  // tokens of 1..4 bytes (empty when a rank exceeds TK_SHORT_MAX_RANK)
    uint32_t short_mask = 0, short_shift = 0;
    std::vector<TkPieceSlot> mid_tab;    // tokens of 5..8 bytes (1..8 without a short table)
    uint32_t mid_mask = 0, mid_shift = 0;
    std::vector<TkPieceSlot> piece;      // tokens of more than 8 bytes
    std::vector<uint32_t> piece_off;
    uint64_t piece_mask = 0;
    std::vector<TkXlSlot> xl;            // tokens of TK_XL_MIN..TK_XL_MAX bytes once more, by identity (tk_common.h)
    uint32_t xl_mask = 0;
    std::vector<uint32_t> xfilter;       // which identity hashes tokens of more than TK_XL_MAX bytes have (tk_common.h)
    double probes_short = 0, probes_mid = 0, probes_long = 0, probes_xl = 0;  // average slots inspected per stored token (build statistics)
    std::vector<TkPairSlot> pair;   // wide format (empty when packed)
    std::vector<uint64_t> pair8;    // packed format (empty when wide)
    uint64_t pair_mask = 0;
    uint64_t n_pairs = 0;
    std::vector<uint32_t> pair2;
    uint32_t byte_rank[256];
    std::vector<uint8_t> spec_bytes;
    std::vector<uint32_t> spec_off, spec_id;
    uint32_t spec_first[8];
    // decoder side (src/lib.rs:323-324): rank -> (offset into tok_bytes / spec_bytes, length).  Ranks are dense in every real vocabulary
    // (0 .. n-1 with a few holes): a direct table `dec_dense[rank]` then; a hash map only when the largest rank is far above the count.
    std::vector<std::pair<uint32_t, uint32_t>> dec_dense;  // length 0 = no such rank
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> dec_sparse, spec_decoder;
    bool dec_is_dense = true;
    uint32_t max_rank = 0;
    const std::pair<uint32_t, uint32_t>* find_token(uint32_t rank) const {
        if (dec_is_dense) return rank < dec_dense.size() && dec_dense[rank].second ? &dec_dense[rank] : nullptr;
        const auto it = dec_sparse.find(rank);
        return it == dec_sparse.end() ? nullptr : &it->second;
    }
    template <class F>
    void for_each_token(F&& f) const {  // f(rank, offset, length)
        if (dec_is_dense) {
            for (size_t r = 0; r < dec_dense.size(); ++r)
                if (dec_dense[r].second) f((uint32_t)r, dec_dense[r].first, dec_dense[r].second);
        } else {
            for (const auto& kv : dec_sparse) f(kv.first, kv.second.first, kv.second.second);
        }
    }
    // ranks ordered by token bytes (token_byte_values, lib.rs:648-650): needed by one API call only, so sorted on first use
    const std::vector<uint32_t>& sorted_ranks() const;
    mutable std::vector<uint32_t> sorted_ranks_;
    mutable std::once_flag sorted_once_;
    uint32_t max_token_len = 0;
    uint64_t n_ranks = 0;

    // exact host lookups through the same tables the device uses
    uint32_t lookup_piece(const uint8_t* p, uint32_t len) const;
    uint32_t lookup_pair(uint32_t a, uint32_t b) const;
};

// key of a byte string as stored in the piece table
uint64_t tk_key_of_bytes(const uint8_t* p, uint32_t len);

// Returns "" on success or an error message.  pat_str: a member of the three scanner families (tk_pattern.cpp) or anything the generic
// engine compiles (tk_regex.cpp).
std::string tk_build_tables(const uint8_t* ranks_blob, const uint64_t* ranks_off, const uint32_t* ranks_ids,
                            uint64_t n_ranks, const uint8_t* spec_blob, const uint64_t* spec_off,
                            const uint32_t* spec_ids, uint64_t n_spec, const char* pat_str, TkHostTables* out);

// family of a pat_str (TK_PAT_*), 3 when it runs on the generic engine, or -1 when it is not supported at all (exported through the
// C ABI, include/tiktoken_amd.h)
extern "C" int tk_pattern_id(const char* pat_str);
// pat_str -> scanner family + parameters, or the generic engine's program: "" or why neither understands the pattern
std::string tk_compile_pattern(const char* pat_str, TkPat* pat, uint16_t* cert, TkRxCompiled* rx);
// the pattern the front kernel runs when the generic engine has done the split: pieces end at hard starts and nowhere else
TkPat tk_nosplit_pat();
// pat_str -> TkPat (tk_pattern.cpp): "" or the reason why the pattern is not supported
// cert_out (may be null): the table of certain piece starts, [16] class masks -- the family's for a stock pattern, else the family's
// minus every pair for which a counter-example exists among all short strings over class representatives and random longer ones
std::string tk_parse_pattern(const char* pat_str, TkPat* out, uint16_t* cert_out = nullptr);
void tk_derive_certain(const TkPat& pat, uint16_t* cert);

// `.tiktoken` text -> packed token bytes + offsets + ranks (reference tiktoken/load.py:159-171).  "" or an error message.
std::string tk_parse_tiktoken(const uint8_t* text, uint64_t len, std::vector<uint8_t>* blob, std::vector<uint64_t>* off,
                              std::vector<uint32_t>* ids);
