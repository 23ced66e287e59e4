/* CPU parity oracle (C) for the BPE encode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm (citations into /root/reference):
 *   src/lib.rs:140-196   _byte_pair_merge        -> tko_bpe_small()
 *   src/lib.rs:47-138    _byte_pair_merge_large  -> tko_bpe_large()
 *   src/lib.rs:198-211   byte_pair_encode        -> tko_encode_piece()
 *   src/lib.rs:360-373   CoreBPE::encode_ordinary-> tko_encode_ordinary()
 *   src/lib.rs:375-442   CoreBPE::encode         -> tko_encode()
 *   tiktoken/core.py:164-176 encode_ordinary_batch (thread pool over documents)
 *                                               -> tko_encode_batch()
 * The regex split (fancy-regex `find_iter`, src/lib.rs:365, a third-party crate that is not
 * under /root/reference: fancy-regex 0.19 / regex 1.13, Cargo.toml:23-24) is restated as a
 * sequential scanner for the three stock patterns of tiktoken_ext/openai_public.py:12-14,89,
 * 104-114; it is pinned against Python `regex.findall(pat_str)` (the substitution the reference
 * itself makes at tiktoken/core.py:395-404) by tests/test_oracle.py and the golden fixtures
 * in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported baseline.  The product (tiktoken_amd/) never links it.
 *
 * Parity status: pinned against the reference's Python layer (tiktoken/_educational.py
 * bpe_encode + regex.findall) and the vocab-free vectors of src/lib.rs:685-701; the reference's
 * real-vocabulary known answers (tests/test_encoding.py) need vocabulary files that are not
 * available offline and run automatically when they are present in $TIKTOKEN_CACHE_DIR.
 */
#ifndef TK_ORACLE_H
#define TK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TKO_PAT_R50K = 0, TKO_PAT_CL100K = 1, TKO_PAT_O200K = 2 };

typedef struct tko_vocab tko_vocab;

/* ranks: n byte strings (blob[off[i]..off[i+1])) with ids[i]; specials likewise (UTF-8). */
tko_vocab* tko_vocab_new(const uint8_t* blob, const uint64_t* off, const uint32_t* ids, uint64_t n,
                         const uint8_t* sblob, const uint64_t* soff, const uint32_t* sids, uint64_t ns,
                         int pattern);
void tko_vocab_free(tko_vocab* v);

/* Pre-tokenise: writes the exclusive end offset of every piece; returns the piece count
 * (or -(needed) if cap is too small). */
int64_t tko_split(int pattern, const uint8_t* text, uint64_t len, uint64_t* piece_ends, uint64_t cap);

/* Encode one piece without regex splitting (src/py.rs:145-150). Returns token count. */
int64_t tko_encode_piece(const tko_vocab* v, const uint8_t* piece, uint64_t len, uint32_t* out, uint64_t cap);

/* Returns the token count, or -1 if cap is too small (cap >= len always suffices). */
int64_t tko_encode_ordinary(const tko_vocab* v, const uint8_t* text, uint64_t len, uint32_t* out, uint64_t cap);

/* allowed_ids: ids of the special tokens that may be emitted (others are encoded as text). */
int64_t tko_encode(const tko_vocab* v, const uint8_t* text, uint64_t len, const uint32_t* allowed_ids,
                   uint64_t n_allowed, uint32_t* out, uint64_t cap);

/* Thread-pool batch (core.py:164-206).  tokens_out must hold doc_off[n_docs] entries; the tokens
 * of document d are written packed, tok_off_out[d]..tok_off_out[d+1].  mode 0 = ordinary,
 * 1 = with allowed specials.  Returns 0. */
int tko_encode_batch(const tko_vocab* v, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs,
                     int mode, const uint32_t* allowed_ids, uint64_t n_allowed, int n_threads,
                     uint32_t* tokens_out, uint64_t* tok_off_out);

/* seconds spent in the parallel per-document encode of the last tko_encode_batch call (the packing pass that follows is excluded) */
double tko_last_encode_seconds(void);

#ifdef __cplusplus
}
#endif

#endif
