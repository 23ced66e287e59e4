// Shared host/device definitions for the MI355X BPE encode path.
//
// Table layouts live in HBM and are read-only after tk_create(); they replace the reference's
// `encoder: FxHashMap<Vec<u8>, Rank>` (src/lib.rs:321) with three exact structures:
//   * piece tables : the whole-piece probe of src/lib.rs:367-368, keyed by the piece's BYTES and split by length so that each
//                    length class runs its own short code path on tables that fit the L2:
//                      short (1..4 bytes): 8-byte slots {key32, rank | (len-1) << 30}, the bytes packed little-endian;
//                      mid   (5..8 bytes): 16-byte slots {key64, rank, len}, the bytes packed little-endian;
//                      long  (> 8 bytes) : 16-byte slots {hash64, rank, len}, verified against the token-bytes blob,
//                    so a hit is always an exact byte match (never a fingerprint alone).
//   * pair table   : open-addressed {(id_left << 32) | id_right -> id_merged} for every vocabulary
//                    token T and every split T = A || B with A and B both vocabulary tokens.  Every
//                    part that ever exists during _byte_pair_merge (src/lib.rs:140-196) is itself
//                    a vocabulary token (single bytes are, and a merge only happens when the
//                    concatenation is a key), and concatenation is unique, so probing this table
//                    with the two part ids is exactly `ranks.get(&piece[a..c])` (lib.rs:150,165-167).
//                    Stored packed in 8 bytes per slot when every id fits 21 bits (all stock
//                    vocabularies), which keeps the whole table at 4 MB -- L2-sized.
//   * pair2 table  : direct 65536-entry table for the initial two-byte probes (lib.rs:150).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TK_HD __host__ __device__ __forceinline__
#else
#define TK_HD inline
#endif

#define TK_RANK_MAX 0xFFFFFFFFu

enum { TK_PAT_R50K = 0, TK_PAT_CL100K = 1, TK_PAT_O200K = 2 };

// Character classes (tools/gen_unicode_tables.py).  Stored one byte per TEXT byte; the low
// nibble is the class, TK_F_HARD marks a position where a new haystack begins (document start,
// special-token start, first byte after a special token): look-ahead from the left sees
// end-of-text there, exactly like the slice `&text[start..end]` of src/lib.rs:405.
enum {
    TK_C_CONT = 0,  // UTF-8 continuation byte (or interior of a special token)
    TK_C_NL = 1,
    TK_C_SP = 2,
    TK_C_WSO = 3,
    TK_C_LU = 4,
    TK_C_LL = 5,
    TK_C_LC = 6,
    TK_C_MK = 7,
    TK_C_NU = 8,
    TK_C_AP = 9,
    TK_C_SL = 10,
    TK_C_OT = 11,
    TK_C_END = 12,
    TK_C_SPEC = 13,  // first byte of an allowed special token occurrence
};
#define TK_F_HARD 0x80u
#define TK_CB(c) (1u << (c))
#define TK_M_WS (TK_CB(TK_C_NL) | TK_CB(TK_C_SP) | TK_CB(TK_C_WSO))
#define TK_M_L (TK_CB(TK_C_LU) | TK_CB(TK_C_LL) | TK_CB(TK_C_LC))
#define TK_M_OTHER (TK_CB(TK_C_MK) | TK_CB(TK_C_AP) | TK_CB(TK_C_SL) | TK_CB(TK_C_OT))
#define TK_M_WORD (TK_M_L | TK_CB(TK_C_MK))
#define TK_M_UPPERISH (TK_CB(TK_C_LU) | TK_CB(TK_C_LC) | TK_CB(TK_C_MK))
#define TK_M_LOWERISH (TK_CB(TK_C_LL) | TK_CB(TK_C_LC) | TK_CB(TK_C_MK))

struct TkPieceSlot {  // 16 bytes
    uint64_t key;     // packed bytes (len <= 8) or tk_hash_bytes (len > 8); ~0 = empty
    uint32_t rank;
    uint32_t len;
};
struct TkShortSlot {  // 8 bytes
    uint32_t key;      // packed bytes (len <= 4)
    uint32_t val;      // rank | (len - 1) << 30; 0xFFFFFFFF = empty
};
#define TK_SHORT_EMPTY 0xFFFFFFFFu
#define TK_SHORT_MAX_RANK 0x3FFFFFFFu  // larger ranks: no short table, pieces of <= 4 bytes live in the mid table
// Pieces of 9..23 bytes -- a fifth of the pieces of web text, and three in five of them are NOT tokens -- are looked up by their IDENTITY:
// three words that hold the bytes themselves and the length (tk_ident), so that a slot answers exactly without a look at the token blob
// (the 16-byte slots of `piece` hold a hash; their candidates are verified in the blob: a chain of dependent loads, and a hash over all
// the bytes first).  The front kernel's in-call table of the pieces that are not tokens is keyed by the same identity (tk_fused.h,
// TkMissKey): one identity, one hash, both slots fetched together.
#define TK_XL_MIN 9u
#define TK_XL_MAX 23u
struct alignas(16) TkXlSlot {  // 32 bytes
    uint64_t w0, w1, w2;  // tk_ident of the token's bytes (w2's top byte is the length)
    uint32_t rank;        // TK_RANK_MAX = empty
    uint32_t pad;
};
struct TkPairSlot {  // 16 bytes
    uint64_t key;    // (id_left << 32) | id_right; ~0 = empty
    uint32_t rank;
    uint32_t pad;
};
#define TK_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// A split pattern of the supported family (reference: pat_str, compiled once per Encoding, src/lib.rs:623).  The three stock patterns
// differ in their alternatives (the `family`); within a family a pattern may vary the contraction list after the apostrophe, its case
// sensitivity, the longest digit group (\p{N}{1,k}; \p{N}+ = unbounded) and the suffix set behind a run of "other" chars
// ([\r\n]*, [\r\n/]*, [/]* or none).  The scanners take a TkPat by value: for the stock patterns it is a compile-time constant.
struct TkPat {
    uint32_t w0;     // family (2 bits) | case-insensitive contractions << 2 | suffix set << 3 (1 = \r\n, 2 = '/') | generic << 5 |
                     // \s+$ before the newline rule << 6 | newline rule \s*[\r\n]+ << 7 | longest digit group << 8 (0 = unbounded) |
                     // number of two-letter contractions << 16
    uint32_t c1;     // one-letter contractions ('s 't ...): bit = letter - 'a'
    uint32_t c2[2];  // two-letter contractions, 16 bits each (first << 8 | second), at most four
    TK_HD constexpr int fam() const { return (int)(w0 & 3u); }
    TK_HD constexpr bool ci() const { return (w0 >> 2) & 1u; }
    TK_HD constexpr uint32_t suffix() const { return (w0 >> 3) & 3u; }
    TK_HD constexpr bool generic() const { return (w0 >> 5) & 1u; }
    TK_HD constexpr bool ws_dollar() const { return (w0 >> 6) & 1u; }
    TK_HD constexpr bool nl_rule() const { return (w0 >> 7) & 1u; }
    TK_HD constexpr uint32_t digits() const { return (w0 >> 8) & 0xFFu; }
    TK_HD constexpr uint32_t n2() const { return (w0 >> 16) & 7u; }
    TK_HD constexpr uint32_t two(uint32_t i) const { return (c2[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu; }
    // class mask of the suffix set
    TK_HD constexpr uint32_t suffix_mask() const { return ((suffix() & 1u) ? TK_CB(TK_C_NL) : 0u) | ((suffix() & 2u) ? TK_CB(TK_C_SL) : 0u); }
    TK_HD constexpr bool same_as(const TkPat& o) const { return ((w0 ^ o.w0) & ~32u) == 0u && c1 == o.c1 && c2[0] == o.c2[0] && c2[1] == o.c2[1]; }
};
#define TK_PAT_GENERIC 3  // kernel template argument: family and parameters are read from TkTables::pat at run time
// the stock pattern of a family: 's 't 'm 'd, 'll 've 're; r50k: case-sensitive, \p{N}+, no suffix; cl100k / o200k: (?i), \p{N}{1,3}
TK_HD constexpr TkPat tk_stock_pat(int fam) {
    return TkPat{(uint32_t)fam | (fam != TK_PAT_R50K ? 4u : 0u) | (fam == TK_PAT_CL100K ? 8u : (fam == TK_PAT_O200K ? 24u : 0u)) |
                     (fam != TK_PAT_O200K ? 64u : 0u) | (fam != TK_PAT_R50K ? 128u : 0u) | (fam != TK_PAT_R50K ? (3u << 8) : 0u) | (3u << 16),
                 (1u << ('s' - 'a')) | (1u << ('t' - 'a')) | (1u << ('m' - 'a')) | (1u << ('d' - 'a')),
                 {((uint32_t)'l' << 8 | 'l') | (((uint32_t)'v' << 8 | 'e') << 16), ((uint32_t)'r' << 8 | 'e')}};
}

// Device-resident view of one encoding's tables.
struct TkTables {
    const uint8_t* uc_stage1;   // [0x1100]
    const uint8_t* uc_stage2;   // [nblocks*256]
    const uint8_t* uc_bmp;      // [65536] the class of every code point below U+10000 once more, directly (one load where the two stages are two
                                // dependent ones: the classification of the front kernel waits for a chain of them per wavefront); null on the host
    const uint32_t* byte_tab;   // [256 * 2] per-byte {class planes, flag planes} of the 16-bytes-per-lane classifier (tk_chunk.h)
    const TkShortSlot* short_tab;  // [short_mask+1] tokens of 1..4 bytes (null when some rank exceeds TK_SHORT_MAX_RANK)
    uint32_t short_mask, short_shift;  // slot = (key32 * K) >> short_shift
    const TkPieceSlot* mid_tab;    // [mid_mask+1] tokens of 5..8 bytes (1..8 without a short table)
    uint32_t mid_mask, mid_shift;
    const TkPieceSlot* piece;   // [piece_mask+1] tokens of more than 8 bytes
    const uint32_t* piece_off;  // [piece_mask+1] offset of the slot's key bytes in tok_bytes
    uint64_t piece_mask;
    uint32_t max_token_len;     // longest vocabulary token in bytes: a longer piece cannot be a token (no probe, no hash of its bytes)
    const uint8_t* tok_bytes;   // all token byte strings, concatenated
    const TkPairSlot* pair;     // [pair_mask+1] wide 16-byte slots (only when some id needs more than 21 bits)
    const uint64_t* pair8;      // [(pair_mask+1)*4] packed 8-byte slots in 4-slot buckets: (id_left:21 | id_right:21 | id_merged:22), ~0 = empty
    uint64_t pair_mask;         // wide: slot mask; packed: BUCKET mask
    const uint32_t* pair2;      // [65536]  rank of the 2-byte string (b0, b1) or TK_RANK_MAX
    const uint32_t* byte_rank;  // [256]
    // special tokens (sorted by bytes); spec_first marks possible first bytes
    const uint8_t* spec_bytes;
    const uint32_t* spec_off;  // [n_spec+1]
    const uint32_t* spec_id;   // [n_spec]
    const uint32_t* spec_head; // [4 * n_spec] {first eight bytes (zero-padded), length, offset}: what tk_special_at asks per special token, in ONE
                               // load that depends on nothing (offsets, then first byte, then the bytes were a chain of four per token; null on the host)
    uint32_t n_spec;
    uint32_t spec_first[8];  // 256-bit set of first bytes
    uint32_t spec_fb;        // the first bytes once more, packed, when there are at most four of them (n_spec_fb; all the stock encodings: '<')
    uint32_t n_spec_fb;      // 0xFF: more than four (the bitmap decides)
    uint32_t spec_second[8]; // 256-bit set of second bytes (all bytes when some special token is a single byte): every stock special token starts "<|"
    int pattern;             // family of the split pattern (TK_PAT_*)
    TkPat pat;               // the pattern itself
    uint16_t cert[16];       // certain piece starts: cert[a] = classes that always start a piece after a char of class a
                             // (the family's table for the stock patterns; derived per pattern otherwise: tk_pattern.cpp)
    const struct TkXlSlot* xl;  // [xl_mask+1] tokens of TK_XL_MIN..TK_XL_MAX bytes once more, 32-byte slots that hold the bytes (below)
    uint32_t xl_mask;
    const uint32_t* xfilter;    // [TK_XFILTER_BITS / 32] bit tk_xfilter_bit(hash of the identity) of every token of more than TK_XL_MAX bytes: a
                                // piece whose bit is clear is not such a token (tk_k_bincount asks before it looks one up)
};
#define TK_XFILTER_BITS (1u << 20)
TK_HD uint32_t tk_xfilter_bit(uint64_t ident_hash) { return (uint32_t)(ident_hash >> 32) & (TK_XFILTER_BITS - 1u); }

TK_HD uint64_t tk_mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return x;
}

// Slot index of a piece key.  The front kernel is VALU-bound and a 64-bit multiply costs four quarter-rate 32-bit
// ones, so this is built from 32-bit multiplies only; on the three vocabularies it probes exactly as well as a
// full 64-bit mixer (1.31 slots per hit, 1.81 per miss at load 0.38).
TK_HD uint64_t tk_piece_slot_hash(uint64_t key, uint32_t len) {
    uint32_t x = (uint32_t)key * 0x9E3779B1u;
    x ^= x >> 15;
    x += (uint32_t)(key >> 32) * 0x85EBCA77u + len * 0xC2B2AE3Du;
    x ^= x >> 13;
    x *= 0x27D4EB2Fu;
    x ^= x >> 16;
    return x;
}
// slots of the short / mid tables: multiplicative hashing, top bits (two or six vector instructions)
TK_HD uint32_t tk_short_slot(uint32_t key, uint32_t shift) { return (key * 0x9E3779B1u) >> shift; }
TK_HD uint32_t tk_mid_slot(uint64_t key, uint32_t shift) {
    uint32_t x = (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u;
    x ^= x >> 15;
    return (x * 0xC2B2AE3Du) >> shift;
}
// Bucket of a pair key: 32-bit multiplies only (this sits on the dependent chain of every merge -- two probes per merge, one merge
// after the other; a 64-bit mixer is three 64-bit multiplies = a dozen quarter-rate instructions in a row).
TK_HD uint64_t tk_pair_slot_hash(uint64_t key) {
    uint32_t x = (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

// ---- identity of a piece (the front kernel's exact keys: TkXlSlot, TkMissKey) ----
// the low clamp(r, 0, 8) bytes of x, left-aligned (r <= 0: nothing; r >= 8: x)
TK_HD uint64_t tk_keep_bytes(uint64_t x, int r) {
    const uint32_t t4 = 4u * (8u - (uint32_t)(r < 0 ? 0 : (r > 8 ? 8 : r)));
    return (x << t4) << t4;
}
// Identity of the piece of `len` bytes whose eight bytes at offset o are ld8(o) (bytes behind the piece may be anything; ld8(16) is
// called for pieces of at most TK_XL_MAX bytes only).  Up to TK_XL_MAX bytes the identity IS the piece: w0, w1 = bytes 0..15, w2 = bytes
// 16..22 below the length in the top byte (each word's bytes left-aligned, the rest zero).  Longer pieces: the first and the last eight
// bytes, and 1 << 63 | length << 32 | `where` (the piece's place in the text, for whoever has to compare the rest).
template <class Ld8>
TK_HD void tk_ident(Ld8&& ld8, uint32_t len, uint32_t where, uint64_t& w0, uint64_t& w1, uint64_t& w2) {
    const bool exact = len <= TK_XL_MAX;
    const uint64_t x0 = ld8(0u), x1 = ld8(exact ? 8u : len - 8u);
    w0 = tk_keep_bytes(x0, (int)len);
    w1 = exact ? tk_keep_bytes(x1, (int)len - 8) : x1;
    w2 = (1ull << 63) | ((uint64_t)len << 32) | where;
    if (exact) w2 = (tk_keep_bytes(ld8(16u), (int)len - 16) >> 8) | ((uint64_t)len << 56);
}
// two 32-bit hashes of an identity (equal identities are compared word for word anyway): the low word selects the slot of the
// vocabulary's table (TkTables::xl), both the slot of the in-call table
TK_HD uint64_t tk_ident_hash(uint64_t w0, uint64_t w1, uint64_t w2, bool exact) {
    const uint32_t a = (uint32_t)w0, b = (uint32_t)(w0 >> 32), c = (uint32_t)w1, d = (uint32_t)(w1 >> 32);
    const uint32_t e = exact ? (uint32_t)w2 : 0u, f = (uint32_t)(w2 >> 32);  // (where a long piece stands is not part of what it is)
    uint32_t h1 = a * 0x9E3779B1u + b * 0x85EBCA77u + c * 0xC2B2AE3Du + d * 0x27D4EB2Fu + e * 0x165667B1u + f * 0xD3A2646Du;
    h1 ^= h1 >> 15;
    h1 *= 0x2C1B3C6Du;
    h1 ^= h1 >> 12;
    uint32_t h2 = (a ^ 0x5BD1E995u) * 0x7FEB352Du + (b + d) * 0x846CA68Bu + (c ^ f) * 0xFD7046C5u + (e + ((a >> 19) | (a << 13))) * 0xB55A4F09u;
    h2 ^= h2 >> 13;
    h2 *= 0x9E3779B1u;
    h2 ^= h2 >> 16;
    return ((uint64_t)h2 << 32) | h1;
}

// streaming hash for keys longer than 8 bytes: fold 8-byte little-endian words (last one zero padded)
TK_HD uint64_t tk_hash_step(uint64_t h, uint64_t w) {
    uint64_t x = (h ^ w) * 0x9FB21C651E98DF25ull;  // one multiply per 8 bytes: equal keys are byte-verified anyway
    x ^= x >> 29;
    return x + 0x9E3779B97F4A7C15ull;
}
#define TK_HASH_SEED 0x243F6A8885A308D3ull
// Keys of pieces longer than TK_KEY_SAMPLED bytes hash the length and four 8-byte words -- the first sixteen and the last sixteen bytes --
// instead of every byte: a key only selects the slots to look at (vocabulary entries are verified against the token blob, missed pieces
// against the claimant's text), and on the device a row of 64 pieces pays for the hash of its LONGEST piece (25 instructions per 8 bytes:
// 50 rounds for a 400-byte run of Thai letters).  The three forms of the function (tk_key_of_bytes on the host, tk_key_of_text, tk_key_of_lds)
// follow this one rule.
#define TK_KEY_SAMPLED 32u
#define TK_PAIR8_ID_BITS 21
#define TK_PAIR8_MAX_ID ((1u << TK_PAIR8_ID_BITS) - 2u)  // ids above this force the wide format
