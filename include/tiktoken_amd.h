/* C ABI of libtiktoken_amd.so -- the MI355X-native replacement for the reference's Rust
 * `_tiktoken` extension on the BPE encode path.
 *
 * Each entry point names the reference interface it replaces (paths into openai/tiktoken v0.14.0).
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md; the Python shim
 * that ships here is tiktoken_amd/_tiktoken.py (ctypes).  All pointers are plain host pointers
 * unless the name says `_device`; no torch/HIP types appear in any signature.
 *
 * Status codes: every function returning int returns one of TK_OK ... TK_UNSUPPORTED; on failure
 * tk_last_error() holds a message (thread-local).  The Python shim maps them to the reference's
 * exception types: TK_VALUE_ERROR -> ValueError (src/py.rs:21-22,46), TK_KEY_ERROR -> KeyError
 * (src/py.rs:142,160,171), TK_RUNTIME_ERROR -> RuntimeError (HIP failures; no reference analogue).
 *
 * There is no CPU implementation behind these calls: without a usable HIP device tk_create fails
 * with TK_RUNTIME_ERROR.
 */
#ifndef TIKTOKEN_AMD_H
#define TIKTOKEN_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TK_OK = 0, TK_VALUE_ERROR = 1, TK_KEY_ERROR = 2, TK_RUNTIME_ERROR = 3, TK_UNSUPPORTED = 4 };

typedef struct tk_core tk_core;

const char* tk_last_error(void);
int tk_device_count(void);

/* CoreBPE.__new__(encoder, special_tokens_encoder, pattern)            src/py.rs:15-23, src/lib.rs:618-663
 * ranks: n_ranks byte strings ranks_blob[ranks_off[i]..ranks_off[i+1]) with ids ranks_ids[i];
 * specials likewise (UTF-8).  pat_str: the split regex, compiled once here (Regex::new, src/lib.rs:623).  The three families of
 * tiktoken_ext/openai_public.py:12-14,89,104-114 (stock strings, their other spellings, variations of the contraction list, digit
 * group, suffix set and white-space rules) run on hand-written scanners; any other pattern in the syntax fancy-regex shares with
 * Python `regex` -- classes (with && and --, POSIX classes), \p{General_Category}, \p{Script}, the binary properties of the UCD, alternation,
 * groups, (?i: ), greedy / lazy / possessive quantifiers, atomic groups, look-ahead, look-behind of fixed length, \b, ^ $ -- is compiled
 * for the generic GPU engine (tk_regex.cpp): into a DFA (leftmost-first; tk_regex_dfa.inc) that every lane walks out of LDS, and -- for
 * what a table cannot express: look-around of several chars, atomic groups and possessive repeats around groups -- into a backtracking
 * program that the GPU interprets ($TIKTOKEN_AMD_RX_MATCHER=program forces the program for every pattern).  Refused with TK_UNSUPPORTED and the reason: look-behind of variable length,
 * back-references (fancy-regex has them; they stay refused here), a pattern that can match the empty string.  Text a pattern does not
 * match yields no tokens, as the reference's find_iter skips it (src/lib.rs:365,405).
 * Text must be valid UTF-8 (the reference's boundary is &str); other bytes never crash but their split is unspecified: a caller that
 * cannot vouch for its bytes checks them with tk_validate_utf8 first.
 * Duplicate ranks -> TK_VALUE_ERROR (the reference panics, src/lib.rs:636-641).  device = HIP device ordinal.
 * Threads: a core may be used from several threads (src/lib.rs:232-238).  Calls on one document of at most 128 KiB without special tokens take
 * no lock (seventy-two slots per core; callers that arrive together share one launch); every other call on a core is serialised by the core's mutex.
 * What a call costs: one kernel launch for a document of up to 128 KiB without special tokens (~22 us for 11 bytes, ~70 us for 4 KiB, ~110 us
 * for 64 KiB: cut at certain piece starts into segments, a workgroup each), the general pipeline's dozen launches otherwise (~0.15 ms and up):
 * the path is for batches. */
int tk_create(const uint8_t* ranks_blob, const uint64_t* ranks_off, const uint32_t* ranks_ids, uint64_t n_ranks,
              const uint8_t* spec_blob, const uint64_t* spec_off, const uint32_t* spec_ids, uint64_t n_spec,
              const char* pat_str, int device, tk_core** out);
void tk_destroy(tk_core* core);

/* How a pat_str would run: 0 r50k/gpt2, 1 cl100k, 2 o200k -- a family the library has hand-written scanners for (the stock patterns,
 * their other spellings, and variations of the contraction list, digit group, suffix set and white-space rules: tk_pattern.cpp);
 * 3 -- the generic engine (tk_regex.cpp); -1 -- not supported (tk_create would say why).  Reference: the regex compiled once per
 * Encoding, src/lib.rs:623. */
int tk_pattern_id(const char* pat_str);

/* Encoding.encode_ordinary_batch / encode_batch                        tiktoken/core.py:164-206
 * == ThreadPool map of CoreBPE.encode_ordinary / encode                src/py.rs:29-49, src/lib.rs:360-442
 * Documents are packed back to back in `utf8`; doc_off has n_docs+1 entries (doc_off[0] == 0).
 * use_special = 0: encode_ordinary semantics.  use_special = 1: special tokens whose id is in
 * allowed_ids[0..n_allowed) are emitted as their id, all other text is ordinary text.
 * On success *tokens_out (library-owned, release with tk_free) holds the token ids of all
 * documents back to back and tok_off_out[0..n_docs] (caller-owned array) their boundaries. */
int tk_encode_batch(tk_core* core, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                    const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** tokens_out, uint64_t* n_tokens_out,
                    uint64_t* tok_off_out);

/* Same, with inputs already resident in HBM and results left in HBM (bench / torch interop).
 * d_utf8 must be readable for 64 bytes past n_bytes.  d_doc_off: uint64[n_docs+1] on the device,
 * h_doc_off: the same offsets on the host (needed only when n_bytes exceeds the per-launch chunk,
 * may be NULL otherwise).  stream: a hipStream_t (NULL = the library's own stream).  The returned
 * device pointers are owned by the core and stay valid until its next encode call. */
int tk_encode_batch_device(tk_core* core, const void* d_utf8, uint64_t n_bytes, const void* d_doc_off,
                           const uint64_t* h_doc_off, uint64_t n_docs, int use_special, const uint32_t* allowed_ids,
                           uint64_t n_allowed, void* stream, const uint32_t** d_tokens_out, uint64_t* n_tokens_out,
                           const uint64_t** d_tok_off_out);

/* Test / debug entry with no reference counterpart: the piece boundaries the GPU pre-tokeniser
 * finds, i.e. what `regex.find_iter` yields at src/lib.rs:365 and :405.  *starts_out receives
 * n_pieces+1 ascending uint32 offsets (last = total bytes); release with tk_free.  Single chunk, less
 * than 2 GiB (bit 31 of an offset marks a char a generic pat_str does not match). */
int tk_pretokenize_batch(tk_core* core, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                         const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** starts_out, uint64_t* n_out);

/* CoreBPE.encode_ordinary(text) / CoreBPE.encode(text, allowed_special)  src/py.rs:29-49
 * (also backs encode_to_tiktoken_buffer, src/py.rs:51-70: the result is a plain uint32 buffer). */
int tk_encode_ordinary(tk_core* core, const uint8_t* utf8, uint64_t len, uint32_t** tokens_out, uint64_t* n_tokens_out);
int tk_encode(tk_core* core, const uint8_t* utf8, uint64_t len, const uint32_t* allowed_ids, uint64_t n_allowed,
              uint32_t** tokens_out, uint64_t* n_tokens_out);

/* CoreBPE.encode_single_piece(piece): BPE of raw bytes without regex splitting   src/py.rs:145-150 */
int tk_encode_single_piece(tk_core* core, const uint8_t* piece, uint64_t len, uint32_t** tokens_out,
                           uint64_t* n_tokens_out);
/* byte_pair_encode(piece, &self.encoder) proper: the merge WITHOUT the whole-piece shortcut (single bytes map to their
 * rank).  What `_encode_unstable_native` calls on re-split candidates      src/lib.rs:198-211, call sites :555,:585-590 */
int tk_byte_pair_encode(tk_core* core, const uint8_t* piece, uint64_t len, uint32_t** tokens_out, uint64_t* n_tokens_out);
/* CoreBPE.encode_single_token(piece)   TK_KEY_ERROR if absent                     src/py.rs:133-143 */
int tk_encode_single_token(tk_core* core, const uint8_t* piece, uint64_t len, uint32_t* token_out);
/* CoreBPE.decode_bytes(tokens)         TK_KEY_ERROR "Invalid token for decoding: N"  src/py.rs:156-162, lib.rs:345-358 */
int tk_decode_bytes(tk_core* core, const uint32_t* tokens, uint64_t n, uint8_t** bytes_out, uint64_t* len_out);
/* The same for a packed batch, on the device: tokens of all documents back to back, tok_off[n_docs + 1] token offsets.  One call
 * replaces the per-document thread pool of Encoding.decode_bytes_batch / decode_batch (tiktoken/core.py:331-350).  *bytes_out:
 * library-owned (tk_free), byte_off_out (may be null): n_docs + 1 byte offsets.  TK_KEY_ERROR as tk_decode_bytes; TK_UNSUPPORTED
 * when the ids are too sparse for a direct table (>= 2^26). */
int tk_decode_batch(tk_core* core, const uint32_t* tokens, const uint64_t* tok_off, uint64_t n_docs, uint8_t** bytes_out,
                    uint64_t* n_bytes_out, uint64_t* byte_off_out);
/* The same with everything resident in HBM: ids (uint32) and token offsets (uint64[n_docs + 1], may be null) on the core's device in,
 * *d_bytes_out / *d_byte_off_out (library-owned device buffers, valid until the core's next decode call; *d_byte_off_out null without
 * d_tok_off) and *n_bytes_out out.  `stream`: a hipStream_t or null (the core's).  The counterpart of tk_encode_batch_device for a
 * consumer that keeps text on the device; no reference counterpart (the reference returns owned Vec<u8>s, src/lib.rs:345-358). */
int tk_decode_batch_device(tk_core* core, const void* d_tokens, uint64_t n_tokens, const void* d_tok_off, uint64_t n_docs, void* stream,
                           const uint8_t** d_bytes_out, uint64_t* n_bytes_out, const uint64_t** d_byte_off_out);
/* CoreBPE.decode_single_token_bytes(token)  (pointer into the core; do not free)  src/py.rs:164-172 */
int tk_decode_single_token_bytes(tk_core* core, uint32_t token, const uint8_t** bytes_out, uint64_t* len_out);
/* CoreBPE.token_byte_values(): tokens in lexicographic byte order                  src/py.rs:178-183, lib.rs:648-650 */
uint64_t tk_n_tokens(tk_core* core);
int tk_sorted_token(tk_core* core, uint64_t i, const uint8_t** bytes_out, uint64_t* len_out, uint32_t* rank_out);
/* The same list in one call: packed bytes + n+1 offsets, owned by the core (valid until tk_destroy). */
int tk_sorted_tokens_packed(tk_core* core, const uint8_t** blob_out, const uint64_t** off_out, uint64_t* n_out);

/* Several GPUs of one node in one process: the batch is split by documents into contiguous ranges of about equal byte counts, every
 * core (one per device; created with tk_create) encodes its range from its own host thread, the token ids come back in document
 * order.  Replaces nothing in the reference (its scaling knob is the thread pool of tiktoken/core.py:175); BASELINE north_star:
 * "inputs shard by document across the GPUs of one node".  The group does not own the cores. */
typedef struct tk_group tk_group;
int tk_group_create(tk_core** cores, uint32_t n, tk_group** out);
void tk_group_destroy(tk_group* group);
uint32_t tk_group_size(tk_group* group);
/* same contract as tk_encode_batch */
int tk_group_encode_batch(tk_group* group, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                          const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** tokens_out, uint64_t* n_tokens_out,
                          uint64_t* tok_off_out);
/* Results gathered on the first core's device; only the text crosses PCIe (every shard's ids stay on its device until the gather).
 * The gather is ONE exchange over xGMI: grouped ncclSend / ncclRecv through RCCL (rccl.h:700-722, loaded on first use) when the
 * group's devices are pairwise distinct, concurrent peer copies (one stream per source device) when a device is named twice or
 * RCCL cannot be loaded ($TIKTOKEN_AMD_NO_RCCL forces the latter).  The pointers are owned by the group (valid until its next call). */
int tk_group_encode_batch_device(tk_group* group, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                                 const uint32_t* allowed_ids, uint64_t n_allowed, const uint32_t** d_tokens_out, uint64_t* n_tokens_out,
                                 const uint64_t** d_tok_off_out);
/* "gathers_rccl" / "gathers_peer": how many device gathers of this group ran on RCCL / on peer copies */
uint64_t tk_group_stat(tk_group* group, const char* name);

/* Vocabulary wire format: the text of a `.tiktoken` file (`base64(token) SP rank` per line) -> the packed arrays tk_create takes.
 * Replaces the per-line Python loop of tiktoken/load.py:159-171, and reads what that loop reads: lines ending in \n, \r or \r\n, two fields
 * between runs of ASCII white space, base64 as b64decode() takes it without validation (bytes outside the alphabet skipped, anything behind
 * the padding ignored), the rank as int() takes it (sign, underscores between digits).  Release the three arrays with tk_free.
 * TK_VALUE_ERROR with "Error parsing line N ..." on malformed input, and on a rank that is negative or does not fit 32 bits (where the
 * reference fails one step later, in CoreBPE's constructor). */
int tk_parse_tiktoken_bpe(const uint8_t* text, uint64_t len, uint8_t** blob_out, uint64_t** off_out, uint32_t** ids_out, uint64_t* n_out);

/* The reference's text boundary is &str -- valid UTF-8 by construction (PyO3 extracts it, src/py.rs:29).  A C caller has bytes: 0 if
 * utf8[0..len) is well-formed UTF-8 (no overlong forms, no surrogates, nothing above U+10FFFF, no truncated char), else TK_VALUE_ERROR
 * with *bad_pos (may be null) = offset of the first byte that is not part of a well-formed char.  Host code; a few GB/s. */
int tk_validate_utf8(const uint8_t* utf8, uint64_t len, uint64_t* bad_pos);

void tk_free(void* p);

/* tk_encode_batch_device alternates between n (1 or 2; default 1) pairs of result buffers: with 2 the ids and offsets of call k stay valid
 * while call k + 1 runs, so a consumer on another stream -- the RCCL send of a shard's ids to the root rank (tiktoken_amd/distributed.py) --
 * overlaps the next encode without a copy of its own.  No reference counterpart (the reference returns owned Vec<u32>s, src/lib.rs:360). */
int tk_set_output_buffers(tk_core* core, uint32_t n);

/* Instrumentation for bench.py: when enabled, every kernel launch of the next encode call is
 * bracketed by HIP events on the stream it runs on; tk_get_kernel_ms returns the summed
 * duration and launch count per kernel name since the last tk_reset_kernel_ms. */
void tk_set_profiling(tk_core* core, int enabled);
void tk_reset_kernel_ms(tk_core* core);
int tk_get_kernel_ms(tk_core* core, const char* kernel_name, double* ms_out, uint64_t* launches_out);
/* Pieces / tokens / bytes handled by the last encode call (for roofline accounting). */
void tk_last_stats(tk_core* core, uint64_t* n_bytes, uint64_t* n_pieces, uint64_t* n_tokens, uint64_t* n_docs,
                   uint64_t* n_medium, uint64_t* n_long);
/* One named figure of the last encode call; 0 for a name it does not know: "chunks", "small_launches", "small_calls", "mid_calls",
 * "back_streams", "regrown", "workspace_bytes", "front_wgs_per_cu", "compute_units". */
uint64_t tk_stat(tk_core* core, const char* name);

#ifdef __cplusplus
}
#endif
#endif
