// The encode pipeline (one launch sequence per chunk of packed documents).  Hot path of
// Encoding.encode_ordinary_batch / encode_batch (tiktoken/core.py:164-206 -> src/lib.rs:360-457).
//
//   tk_k_front<PAT,SPEC,MODE>  per 4 KiB window (a tile of 3840 bytes + 128 of context on either side): classify every byte, build the
//                         class-set bitmaps, find the piece starts (regex pre-tokenisation, src/lib.rs:365), enumerate the pieces that
//                         START in the tile and probe each one whole in the vocabulary (src/lib.rs:367) from the LDS copy of the text.
//                         Results form a run at piece id  pid = tile * 4096 + k.  A piece that is not a token claims a slot of the
//                         in-call table of missed pieces, or finds it claimed by identical bytes (exact: the slot holds the bytes of
//                         pieces of up to 23 bytes, longer ones are compared in the text); its result word refers to that slot either way
//                         (TkMissData).  No piece offsets travel through HBM.  Three instances (TKF_MODE_*): one workgroup per tile;
//                         the deferred tiles' piece starts by the workgroup-wide scanner; the rest of the deferred tiles from given starts.
//   tk_k_bincount + tk_k_scan_small + tk_k_binfill   occupied slots -> length-binned lists, without global atomics
//   tk_k_merge_all       every piece of 2..1024 bytes: byte_pair_merge in LDS, 1..64 lanes per piece (src/lib.rs:140-196)
//   tk_k_merge_llane<N>, tk_k_merge_group<G>   the same, a kernel per length bin (vocabularies with ids above 2^21)
//   tk_k_merge_rounds / _long   longer pieces: in rounds, or one merge at a time over a 64-ary min tree (same result as lib.rs:47-138)
//   tk_k_count_tiles      token count per tile (a missed piece's count from its entry)
//   tk_k_scan_*           exclusive scan of the tile counts
//   tk_k_place            per tile: tokens to their final place (a missed piece's tokens from its entry), whole lines through LDS
//   tk_k_docoff          per document: token offset of the piece that starts it
//   tk_k_small            one document of up to 2 KiB (or one segment of a mid-size document) by one workgroup, start to finish
//
// Tile rule (checked on the CPU by tests/test_device_logic_sim.py): a tile derives exactly the piece starts
// inside its own byte range.  Scanners start at the tile's certain starts plus the last certain start before the
// tile (128 bytes of left context, else the tile is deferred to the workgroup-wide scanner), record only boundaries
// inside the tile and stop at its end.  Nothing crosses tiles, so tiles are fully independent.
//
// Three measured facts shape the code: (1) tk_k_front runs at the hardware's eight wavefronts per SIMD and its time is the
// time a workgroup needs for its tile -- 80 000 cycles, of which its own vector instructions are a ninth; the rest is waiting for its turn
// on the SIMD (eight wavefronts share it, the vector ALU is busy 80 % of the time) and for memory.  What shortens it is fewer
// instructions ON THE LONGEST WAVEFRONT'S PATH of every phase and fewer dependent memory round trips (-DTKF_TIMING: cycles per phase,
// profiles/r06_front_cycles.txt): an instruction moved onto one wavefront while three wait at a barrier is not saved.  (2) The classification
// loop is written with integer flags instead of short-circuit control flow (a wave64 VALU instruction occupies its SIMD for four cycles
// whatever the lanes do).  (3) Same-address returning atomics run at only 10..130 M/s on this multi-XCD part, so nothing on the path
// allocates through a global counter.
#pragma once
#include <type_traits>

#include "tk_chunk.h"
#include "tk_kernels.h"

#define TKF_NONE 0xFFFFFFFFu
#define TK_MERGE_REDO 0xFFFFFFFFu  // TkMissData::res_cnt after tk_k_merge_rounds: the piece has to go through tk_k_merge_long
#define TK_BIGCOPY 4096      // token runs from this length on are copied by tk_k_bigcopy
#define TK_BIGCOPY_CAP 1024  // entries of its list
#define TKF_CHAIN_END 0xFFFFFFFFFFFFFFFFull
#define TKF_CONT_CAP 768       // continuation list of the scanners: deferred-tile variant (own array)
#define TKF_CONT_CAP_FAST 768  // ... one-workgroup-per-tile variant (the list lives in the byte table's LDS, dead after phase B)
#define TKF_SLOW_CAP 8    // pieces of one tile that leave its window
// A deferred tile in a stretch without certain starts walks towards itself from the stretch's start, a 4 KiB window per step; every tile
// of the stretch does: quadratic.  With TKF_DBG_MAY_GIVE_UP in `dbg` a tile gives up after TKF_WALK_BUDGET windows and goes on a second
// list (behind the first, count in TK_CNT_DEFER2); the host then lets the generic engine split the chunk -- every piece start becomes a
// hard start -- and runs the kernel again over that list (TKF_DBG_SECOND).  TIKTOKEN_AMD_DEBUG bit 0x20000000: a budget of zero windows
// (tests: ordinary corpora through this path).
#define TKF_WALK_BUDGET 32u
#define TKF_DBG_MAY_GIVE_UP 0x8000000
#define TKF_DBG_SECOND 0x10000000
#define TKF_DBG_NO_BUDGET 0x20000000
// Every piece start of the chunk is a hard start already (the generic engine has split it: tk_api.hip, rx_split): the tile's piece starts
// ARE its hard starts, phases B-D (classes, certain starts, scanners: 2 of the kernel's 5.3 ms per GiB) are skipped by the one-tile-per-
// workgroup instances.  The deferred-tile instance never takes this way (it is the general one).
#define TKF_DBG_HARD_ONLY 0x40000000
// The deferred tiles in TWO kernels (round 5).  The scanner of the deferred tiles compiles at 128 registers with ~70 spilled, four
// workgroups per CU: so its instance (TKF_MODE_STARTS) only FINDS the piece starts -- it leaves the tile's start bitmap and, in tile_np,
// where the tile's last piece ends -- and a third instance (TKF_MODE_GIVEN), launched over the deferred list, takes starts and end from
// there and does the rest (phases E and F) at the eight workgroups per CU of the common instance (TKF_MODE_TILE: one workgroup per tile,
// phases A-F).  Three instances, three kernel names in a trace.
#define TKF_MODE_TILE 0
#define TKF_MODE_STARTS 1
#define TKF_MODE_GIVEN 2
#define TKF_BATCH 896  // pieces per part of a tile in the front kernel's phase F (the class lists hold 1024 entries)
#define TKF_BL 14      // ... = the 28 bitmap words = 896 positions of this many lanes of phase E, when a tile has more pieces than that
#define TKF_CAP 4096  // piece ids per tile: pid = tile * TKF_CAP + k (a 4096-byte tile starts at most 4096 pieces)
// The tail of a tile's run of result words holds, from the back: the number of its pieces that are not tokens (TKF_TAIL_NMISS), the number
// of its gap chars (TKF_TAIL_NGAP), then the entries of the pieces that are not tokens once more, in no particular order -- what the counting
// pass of the back end needs, 0.12 GB per GiB instead of all the result words (0.65 GB).  It always fits: a piece that is not a token has at
// least two bytes, so pieces + missed pieces <= the tile's 3840 bytes.
#define TKF_TAIL_NMISS (TKF_CAP - 1)
#define TKF_TAIL_NGAP (TKF_CAP - 2)
#define TKF_TAIL_REFS (TKF_CAP - 3)  // the q-th missed piece's entry: res[run + TKF_TAIL_REFS - q]

// Per-piece result word res[piece id]: a token id (the piece is a vocabulary token, src/lib.rs:367), or a reference to where its
// tokens will be once the merge kernels have run.
#define TK_RES_FLAG 0x80000000u  // not a single token: TK_RES_FLAG | i = entry i of the chunk's miss data (TkMissData: its result is there)
#define TK_RES_GAP 0x7FFFFFFFu   // no token at all: a char at which a pat_str of the generic engine matches nothing (find_iter skips it, src/lib.rs:365)
// Pieces that are not vocabulary tokens ("missed" pieces: they have to be merged, src/lib.rs:369).  A batch repeats them -- 30 M per GiB of
// web text, 1.2 M distinct -- so each DISTINCT one is merged once: the front kernel claims a slot of an open-addressed table keyed by the
// piece's bytes (TkMissKey; exact: equal hashes are verified byte for byte), every occurrence refers to the slot, the merge kernels leave
// the result in it and tk_k_place reads it from there.  The slot index is the index of its TkMissData entry; behind the table's entries
// the array goes on with the OVERFLOW entries, handed out by a counter: pieces the table does not take (longer than TK_GLANE_MAX bytes, a
// full neighbourhood, chunks too small for a table).  No per-tile lists: the merge work list is a walk over this array.
// Round 5: a slot is 32 bytes and holds the claimant's IDENTITY -- for a piece of at most TK_XL_MAX bytes the bytes themselves
// (w0, w1: bytes 0..15, w2: bytes 16..22 and the length in the top byte; each word's bytes left-aligned, the rest zero), so that a later
// occurrence is recognised from the slot alone: no look at the claimant's text (a dependent random line, and a compare loop whose trip
// count was the longest piece's of a row of 64), no hash over all the bytes (25 instructions per 8 bytes).  Longer pieces: w0 / w1 = the
// first / last eight bytes, w2 = 1 << 63 | length << 32 | start, compared in the text as before.  The key word is a hash of the identity
// (without the start); ~0 = empty.  The identity words are ~0 until the claimant has written them (a word of bytes is never ~0 together
// with a valid w2: w2's top byte is a length <= 23, or bit 63 with bits 42..62 clear).
struct TkMissKey {
    unsigned long long key;  // ~0 = empty
    unsigned long long w0, w1, w2;
};
// Where a distinct missed piece and its result live: a table slot has a 64-byte line of its own -- {start, len, count, tokens}: a piece of up
// to TKD_INLINE tokens (97.5 % of the missed pieces of the web-text corpus) has them IN the entry, so an occurrence costs ONE random
// access; an overflow entry is 16 bytes (sized for the worst case, n / 2 of them), its tokens are in the staging area.
// res_cnt: the token count; with TKD_INLINE_BIT the tokens are tok[0 .. count); without it tok[0] (res_tok) is the token (count 1) or
// the staging position of the tokens.  TK_MERGE_REDO: see above.
#define TKD_INLINE 13
#define TKD_INLINE_BIT 0x40000000u
#define TKD_COUNT(w) ((w) & 0x3FFFFFFFu)
struct TkMissTab {  // 64 bytes
    uint32_t start, len, res_cnt, tok[TKD_INLINE];
};
struct TkMissOvf {  // 16 bytes
    uint32_t start, len, res_cnt, res_tok;
};
struct TkMiss {
    TkMissTab* tab;     // [ovf_base] the table's entries (slot index = entry index; written by whoever claims the slot: never cleared)
    TkMissOvf* ovf;     // [..] the overflow entries, index i - ovf_base
    uint32_t ovf_base;  // slots of the table (0 without one)
    uint8_t* cnt8;      // [ovf_base] token count of a table slot's piece once more, one BYTE per slot (255: look at the entry): the counting
                        // pass of the back end needs nothing else of an entry, and four million of these stay in the L2
    __device__ __forceinline__ uint32_t* head(uint32_t i) const { return i < ovf_base ? &tab[i].start : &ovf[i - ovf_base].start; }
    __device__ __forceinline__ uint2 piece(uint32_t i) const { return *(const uint2*)head(i); }                 // {start, len}
    __device__ __forceinline__ uint2 result(uint32_t i) const { return *(const uint2*)(head(i) + 2); }          // {res_cnt, tok[0]}
    __device__ __forceinline__ void put(uint32_t i, uint32_t cnt, uint32_t tok) const {
        *(uint2*)(head(i) + 2) = make_uint2(cnt, tok);
        if (i < ovf_base) cnt8[i] = (uint8_t)(TKD_COUNT(cnt) < 255u ? TKD_COUNT(cnt) : 255u);
    }
    // where the merge kernels write a piece's `total` tokens: into the entry when they fit (then put(i, total | TKD_INLINE_BIT, ..) is
    // only the count word: put_count), else to the staging area at the piece's text position s
    __device__ __forceinline__ bool fits(uint32_t i, uint32_t total) const { return i < ovf_base && total <= (uint32_t)TKD_INLINE; }
    __device__ __forceinline__ void put_count(uint32_t i, uint32_t w) const {
        head(i)[2] = w;
        if (i < ovf_base) cnt8[i] = (uint8_t)(TKD_COUNT(w) < 255u ? TKD_COUNT(w) : 255u);
    }
    __device__ __forceinline__ uint32_t count(uint32_t i) const {  // token count of entry i's piece
        if (i < ovf_base) {
            const uint32_t c = cnt8[i];
            if (c != 255u) return c;
        }
        return TKD_COUNT(head(i)[2]);
    }
};
struct TkFrontOut {
    uint32_t* starts;     // piece-start bitmap (n/32 words; each tile stores its own 120 words)
    uint32_t* tile_np;    // pieces per tile
    uint32_t* res;        // [piece id] see above
    uint8_t* tile_sum;    // per tile, written when a tile is deferred for a walk back: its class if the whole tile is one run of that
                          // class without a certain start (later tiles jump over it), 16 if not; 0xFF: not written
    TkMiss data;          // table slots + overflow entries (data.ovf_base: index of the first overflow entry)
    uint32_t ovf_cap;     // overflow entries there is room for (TK_CNT_OVF counts on beyond it: the host then repeats the batch with more)
    uint32_t* listC;      // {data index, start, len, scratch bytes before, tree levels before} for > 1 KiB pieces
    uint32_t* counters;
};

// ------------------------------------------------------------------------------------------
// Workgroup-wide scanning straight from global memory, for the rare places where a tile's LDS window is not enough: a piece that
// leaves the window (a run of a million letters is ONE piece), and the search for the last certain piece start before a tile whose
// left context has none.  All 256 threads take part (uniform control flow): every lane classifies 16 bytes (tk_chunk.h), a query
// advances 4 KiB per step.  The scanner itself is tk_piece_end_runs (tk_device.h) over these run queries.
// ------------------------------------------------------------------------------------------
struct TkCoop {
    const TkTables* T;
    const uint8_t* text;
    uint64_t n;
    const uint32_t *brk, *ss, *si;
    const uint32_t* btab;  // LDS: byte table
    uint32_t* red;         // LDS [8]
    uint8_t* lastc;        // LDS [256]
    TkPat pat;
};

// masks of the 16 text bytes at g (a multiple of 16, may be negative or beyond the text)
// (one copy: the callers are inlined into the scanner at several places and the instruction cache is small)
__device__ __noinline__ void tk_coop_chunk(const TkCoop& C, int64_t g, TkChunkMasks& mk) {
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t valid = 0, past = 0;
    if (g >= 0 && (uint64_t)g < C.n) {
        const uint4 x = *(const uint4*)(C.text + g);
        w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
        const uint64_t left = C.n - (uint64_t)g;
        valid = left >= 16 ? 0xFFFFu : ((1u << (uint32_t)left) - 1u);
        if (left < 16) {
            const uint32_t nb = (uint32_t)left;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int lo = 4 * d;
                w[d] &= nb >= (uint32_t)lo + 4u ? 0xFFFFFFFFu : (nb <= (uint32_t)lo ? 0u : ((1u << (8u * (nb - lo))) - 1u));
            }
        }
    }
    if (g >= 0) past = ~valid & 0xFFFFu;
    TkChunk ch;
    auto tab = [&](uint32_t b, uint32_t& x, uint32_t& y) {
        const uint2 e = *(const uint2*)&C.btab[b * 2];
        x = e.x;
        y = e.y;
    };
    tk_chunk_table_pass(w, tab, ch);
    // Non-ASCII chars: code point -> class through the two-stage table.  One workgroup walks here step by step, so the latency of a step
    // is what counts: the (at most eight) lead bytes of the chunk are decoded from registers and their table loads are issued together,
    // two dependent loads per chunk instead of two per char.
    const bool has_prev = g >= 16 && valid;
    const uint32_t prev = has_prev ? *(const uint32_t*)(C.text + g - 4) : 0u;
    const uint32_t w4 = (valid && (uint64_t)g + 16 < C.n) ? *(const uint32_t*)(C.text + g + 16) : 0u;
    if (valid && ((w[0] | w[1] | w[2] | w[3]) & 0x80808080u)) {
        auto cls_idx = [&](uint32_t cp) -> uint32_t { return cp > 0x10FFFFu ? 0xFFFFu : cp; };
        const uint32_t cont = tk_plane16(ch.f0, ch.f1, 0);
        uint32_t leads = tk_plane16(ch.f0, ch.f1, 1) | tk_plane16(ch.f0, ch.f1, 2) | tk_plane16(ch.f0, ch.f1, 3);
        int kk[9];
        uint32_t cp[9], ln[9], s1[9], act = 0;
        {  // a char that straddles in from the previous chunk
            const bool on = (cont & 1u) && has_prev;
            const int k = (prev >> 24) >= 0xC0u ? -1 : (((prev >> 16) & 0xFFu) >= 0xC0u ? -2 : -3);
            kk[8] = k;
            cp[8] = cls_idx(tk_utf8_cp(__builtin_amdgcn_alignbyte(w[0], prev, (uint32_t)(4 + k)), &ln[8]));
            act |= on ? 256u : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = leads ? __ffs((int)leads) - 1 : 0;
            act |= leads ? (1u << q) : 0u;
            leads &= leads - 1;
            kk[q] = k;
            const int d = k >> 2;
            const uint32_t lo = d == 0 ? w[0] : (d == 1 ? w[1] : (d == 2 ? w[2] : w[3])), hi = d == 0 ? w[1] : (d == 1 ? w[2] : (d == 2 ? w[3] : w4));
            cp[q] = cls_idx(tk_utf8_cp(__builtin_amdgcn_alignbyte(hi, lo, (uint32_t)k & 3u), &ln[q]));
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) s1[q] = C.T->uc_stage1[cp[q] >> 8];
#pragma unroll
        for (int q = 0; q < 9; ++q) s1[q] = C.T->uc_stage2[s1[q] * 256u + (cp[q] & 255u)];
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if ((act >> q) & 1u) tk_chunk_apply(ch, kk[q], ln[q], s1[q]);
    }
    uint32_t brk16 = 0, ss16 = 0, si16 = 0;
    if (valid) {
        const uint32_t sh = (uint32_t)g & 16u;
        brk16 = (C.brk[g >> 5] >> sh) & 0xFFFFu;
        if (C.ss) {
            ss16 = (C.ss[g >> 5] >> sh) & 0xFFFFu;
            si16 = (C.si[g >> 5] >> sh) & 0xFFFFu;
        }
    }
    tk_chunk_finalize(ch, valid, past, brk16, ss16, si16, mk);
}
// bytes of the chunk whose class nibble is in the class mask cm
__device__ __forceinline__ uint32_t tk_member16(const uint32_t p[4], uint32_t cm) {
    const uint32_t n0 = ~p[0], n1 = ~p[1], n2 = ~p[2], n3 = ~p[3];
    const uint32_t q[4] = {n3 & n2, n3 & p[2], p[3] & n2, p[3] & p[2]}, r[4] = {n1 & n0, n1 & p[0], p[1] & n0, p[1] & p[0]};
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) m |= ((cm >> c) & 1u) ? (q[c >> 2] & r[c & 3]) : 0u;
    return m & 0xFFFFu;
}
// first / last lane's value over the workgroup (v = TKF_NONE where a lane has none; lanes are in position order)
__device__ __forceinline__ uint32_t tk_coop_first(const TkCoop& C, uint32_t v, bool last) {
    const uint64_t m = __ballot(v != TKF_NONE);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t wv = TKF_NONE;
    if (m) wv = (uint32_t)__shfl((int)v, last ? 63 - __clzll((long long)m) : __ffsll((unsigned long long)m) - 1, 64);
    if (lane == 0) C.red[wid] = wv;
    __syncthreads();
    uint32_t res = TKF_NONE;
    if (last) {
        for (int q = 0; q < 4; ++q)
            if (C.red[q] != TKF_NONE) res = C.red[q];
    } else {
        for (int q = 3; q >= 0; --q)
            if (C.red[q] != TKF_NONE) res = C.red[q];
    }
    __syncthreads();
    return res;
}
// run_end of tk_piece_end_runs: first char start s >= from whose look-ahead class (END at hard starts and past the text) is not in cm
__device__ __noinline__ uint64_t tk_coop_run_end(const TkCoop& C, uint64_t from, uint32_t cm) {
    for (uint64_t wb = from & ~15ull;; wb += 4096) {
        const uint64_t g = wb + 16ull * threadIdx.x;
        TkChunkMasks mk;
        tk_coop_chunk(C, (int64_t)g, mk);
        uint32_t stopm = (~tk_member16(mk.p, cm) | mk.hard) & 0xFFFFu;
        if (g + 16 <= from) stopm = 0;
        else if (g < from) stopm &= ~((1u << (uint32_t)(from - g)) - 1u);
        const uint32_t first = stopm ? (uint32_t)(g - wb) + (uint32_t)__ffs((int)stopm) - 1u : TKF_NONE;
        const uint32_t res = tk_coop_first(C, first, false);
        if (res != TKF_NONE) return wb + res;
    }
}
// last_in of tk_piece_end_runs: start of the last char of [from, to) whose class is in cm, or TK_NO_POS
__device__ __noinline__ uint64_t tk_coop_last_in(const TkCoop& C, uint64_t from, uint64_t to, uint32_t cm) {
    if (to <= from) return TK_NO_POS;
    for (int64_t wb = (int64_t)((to - 1) & ~15ull) - 4080;; wb -= 4096) {
        const int64_t g = wb + 16ll * threadIdx.x;
        TkChunkMasks mk;
        tk_coop_chunk(C, g, mk);
        uint32_t cand = tk_member16(mk.p, cm) & mk.text;
        if (g >= (int64_t)to || g + 16 <= (int64_t)from) cand = 0;
        else {
            if (g + 16 > (int64_t)to) cand &= (1u << (uint32_t)((int64_t)to - g)) - 1u;
            if (g < (int64_t)from) cand &= ~((1u << (uint32_t)((int64_t)from - g)) - 1u);
        }
        const uint32_t last = cand ? (uint32_t)(g - wb) + 31u - (uint32_t)__clz((int)cand) : TKF_NONE;
        const uint32_t res = tk_coop_first(C, last, true);
        if (res != TKF_NONE) return (uint64_t)(wb + res);
        if (wb <= (int64_t)from) return TK_NO_POS;
    }
}
// last certain piece start at or before `pos` (exists: position 0 and document starts are hard starts).  *last_other gets the highest
// position in (result, pos] whose class differs from the class of the byte at `pos` (TK_NO_POS: none): when there is none behind the
// first char, everything between the returned start and `pos` is one run of a single class.
// cref_in < 16: the class to compare with instead of the class at `pos` (the caller has skipped a stretch of that class).
__device__ __noinline__ uint64_t tk_coop_certain_before(const TkCoop* Cp, uint64_t pos, uint32_t cref_in, uint64_t* last_other, uint32_t* cls_at_pos) {
    const TkCoop& C = *Cp;
    uint64_t other = TK_NO_POS;
    uint32_t cref = cref_in;  // (16: set in the first span by the lane that covers `pos`)
    for (int64_t wb = (int64_t)(pos & ~15ull) - 4080;; wb -= 4080) {  // (16 bytes of overlap: the first chunk of a span has no known
        const int64_t g = wb + 16ll * threadIdx.x;                    //  predecessor; the next span sees it as its last chunk)
        TkChunkMasks mk;
        tk_coop_chunk(C, g, mk);
        TkSets st;
        tk_sets_from_planes(mk.p[0], mk.p[1], mk.p[2], mk.p[3], st);
        C.lastc[threadIdx.x] = (uint8_t)tk_class_from_planes(mk.p, 15);
        if (cref == 16 && g <= (int64_t)pos && (int64_t)pos < g + 16) C.red[4] = tk_class_from_planes(mk.p, (uint32_t)((int64_t)pos - g));
        __syncthreads();
        if (cref == 16) cref = C.red[4];
        const uint32_t prevc = threadIdx.x ? (uint32_t)C.lastc[threadIdx.x - 1] : 0u;
        uint32_t apb = 7u;  // apostrophes in the three bytes before the chunk
        if (g >= 4 && (uint64_t)g < C.n) {
            const uint32_t pv = *(const uint32_t*)(C.text + g - 4);
            apb = (uint32_t)(((pv >> 8) & 0xFFu) == 0x27u) | ((uint32_t)(((pv >> 16) & 0xFFu) == 0x27u) << 1) | ((uint32_t)((pv >> 24) == 0x27u) << 2);
        }
        uint32_t cert = C.pat.generic() ? tk_chunk_certain_rt(C.T->cert, st, mk.text, mk.hard & mk.text, prevc)
                                        : tk_chunk_certain(C.pat.fam(), st, mk.text, mk.hard & mk.text, prevc, tk_chunk_near(st.ap, apb));
        uint32_t oth = ~tk_member16(mk.p, 1u << cref) & 0xFFFFu;  // bytes of another class
        if (g > (int64_t)pos) cert = oth = 0;
        else if (g + 16 > (int64_t)pos + 1) {
            cert &= (2u << (uint32_t)((int64_t)pos - g)) - 1u;
            oth &= (2u << (uint32_t)((int64_t)pos - g)) - 1u;
        }
        if (g < 0) oth = 0;
        const uint32_t last = cert ? (uint32_t)(g - wb) + 31u - (uint32_t)__clz((int)cert) : TKF_NONE;
        const uint32_t res = tk_coop_first(C, last, true);  // (its barriers also protect lastc and red[4])
        if (other == TK_NO_POS) {
            const uint32_t lo = oth ? (uint32_t)(g - wb) + 31u - (uint32_t)__clz((int)oth) : TKF_NONE;
            const uint32_t ro = tk_coop_first(C, lo, true);
            if (ro != TKF_NONE) other = (uint64_t)(wb + ro);
        }
        if (res != TKF_NONE) {
            const uint64_t p0 = (uint64_t)(wb + res);
            *last_other = (other != TK_NO_POS && other > p0) ? other : TK_NO_POS;
            *cls_at_pos = cref;
            return p0;
        }
    }
}
struct TkCoopAcc {  // provider of tk_piece_end_runs: single positions straight from global memory (uniform loads), runs by the workgroup
    const TkCoop& C;
    __device__ __forceinline__ uint32_t cls(uint64_t pos) const { return pos >= C.n ? (uint32_t)TK_C_END : tk_class_byte_slow(C.T, C.text, pos, C.n, C.brk, C.ss, C.si); }
    __device__ __forceinline__ uint32_t byte(uint64_t pos) const { return C.text[pos]; }
    __device__ __forceinline__ uint64_t run_end(uint64_t from, uint32_t cm) const { return tk_coop_run_end(C, from, cm); }
    __device__ __forceinline__ uint64_t last_in(uint64_t from, uint64_t to, uint32_t cm) const { return tk_coop_last_in(C, from, to, cm); }
};
// end of the piece that starts at p (any length), by the whole workgroup
__device__ __noinline__ uint64_t tk_coop_piece_end(const TkCoop* Cp, uint64_t p) {
    TkCoopAcc acc{*Cp};
    uint64_t e = tk_piece_end_runs(acc, p, Cp->pat);
    if (e <= p) e = tk_next_char(acc, p);
    return e > Cp->n ? Cp->n : e;
}
// A chain of SHORT pieces with uncertain boundaries ("x'llx'll...": whether 'll ends a piece depends on what stands before it, as far
// back as one cares to look) cannot be cut by any local rule; a tile in such a stretch walks from the last certain start, and piece by
// piece with the whole workgroup that is 12 us per piece.  This walks a 4 KiB window at a time instead: classes of the window (16 bytes
// per lane, as everywhere), then every lane evaluates the piece that WOULD start at each of its char starts (tk_piece_end on the LDS
// copy), then pointer doubling follows the chain from p in twelve steps.  Returns the last position of the chain from p whose piece ends
// inside the window and before `target` (p itself if its piece does not: the caller evaluates that one by the run queries).
struct TkWalkLds {
    uint8_t* raw;       // [TK2_WIN + 16]
    uint32_t* planes;   // [4 * TK2_PLW]
    uint32_t* start;    // [TK2_WIN / 32 + 4]
    uint32_t* hard;     // [TK2_WIN / 32 + 4]
    uint16_t* jump;     // [TK2_WIN]
};
struct TkWalkAcc {  // accessor over the walk's window for ONE evaluation: looking past `lim` (window offset) sets `left` -- the evaluation is
                    // then dropped, which also bounds its cost (a window of one letter would otherwise be walked to its end from every byte)
    const uint32_t *planes32, *start32, *hard32;
    const uint8_t* raw;
    int64_t base;
    uint64_t n;
    uint32_t lim;
    bool left;
    __device__ __forceinline__ uint32_t cls(uint64_t pos) {
        if (pos >= n) return TK_C_END;
        const int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < (int64_t)lim) {
            const uint32_t wi = (uint32_t)r >> 5, b = (uint32_t)r & 31u;
            if (!((start32[wi] >> b) & 1u)) return (uint32_t)TK_C_CONT;
            return tk_class_at_lds(planes32, (uint32_t)r) | (((hard32[wi] >> b) & 1u) << 7);
        }
        left = true;
        return TK_C_END;
    }
    __device__ __forceinline__ uint32_t byte(uint64_t pos) {
        const int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < (int64_t)lim) return raw[r];
        left = true;
        return 0;
    }
};
#define TK_WALK_PIECE 256u  // longest piece the window walk follows (longer ones are the business of the run queries)
__device__ __noinline__ uint64_t tk_coop_window_walk(const TkCoop* Cp, const TkWalkLds* Lp, uint64_t p, uint64_t target) {
    const TkCoop& C = *Cp;
    const TkWalkLds& L = *Lp;
    const uint32_t tid = threadIdx.x;
    const uint64_t wb = p & ~15ull;
    const uint64_t g = wb + 16ull * tid;
    TkChunkMasks mk;
    tk_coop_chunk(C, (int64_t)g, mk);
    {
        constexpr uint32_t HW = TK2_PLW * 2u;  // halfwords per plane
        uint16_t* p16 = (uint16_t*)L.planes;
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) p16[pl * HW + tid] = (uint16_t)mk.p[pl];
        ((uint16_t*)L.start)[tid] = (uint16_t)mk.start;
        ((uint16_t*)L.hard)[tid] = (uint16_t)mk.hard;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (g < C.n) x = *(const uint4*)(C.text + g);  // (the text is readable 64 bytes past its end)
        *(uint4*)(L.raw + 16u * tid) = x;
    }
    __syncthreads();
    TkWalkAcc acc{L.planes, L.start, L.hard, L.raw, (int64_t)wb, C.n, (uint32_t)TK2_WIN, false};
    for (uint32_t j = 0; j < 16u; ++j) {
        const uint64_t q = g + j;
        uint32_t v = 16u * tid + j;  // a position the chain cannot leave: no piece start, outside [p, target), or unresolved in this window
        if (((mk.start >> j) & 1u) && q >= p && q < target && q < C.n) {
            acc.left = false;
            acc.lim = v + TK_WALK_PIECE < (uint32_t)TK2_WIN ? v + TK_WALK_PIECE : (uint32_t)TK2_WIN;
            uint64_t e = tk_piece_end(acc, q, C.pat);
            if (e <= q) e = tk_next_char(acc, q);
            if (!acc.left && e < target && e - wb < (uint64_t)TK2_WIN) v = (uint32_t)(e - wb);
        }
        L.jump[16u * tid + j] = (uint16_t)v;
    }
    __syncthreads();
    for (int it = 0; it < 12; ++it) {  // jump <- jump o jump
        uint16_t nv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) nv[j] = L.jump[L.jump[16u * tid + j]];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) L.jump[16u * tid + j] = nv[j];
        __syncthreads();
    }
    const uint32_t t = L.jump[(uint32_t)(p - wb)];
    __syncthreads();
    return wb + t;
}

// In a run of ASCII digits the pieces of the cl100k / o200k patterns are groups of three from the start of the run (\p{N}{1,3}): a chain
// that walks such a run towards `target` can jump over the whole groups (a megabyte of digits would otherwise be 350 000 evaluations).
__device__ __noinline__ uint64_t tk_coop_skip_digit_groups(const TkCoop* Cp, uint64_t p, uint64_t target) {
    const TkCoop& C = *Cp;
    uint64_t r2 = 0;
    for (uint64_t wb = p & ~15ull;; wb += 4096) {  // end of the run of ASCII digits that are not hard starts
        const uint64_t g = wb + 16ull * threadIdx.x;
        TkChunkMasks mk;
        tk_coop_chunk(C, (int64_t)g, mk);
        uint32_t ascii = 0;
        if (g < C.n) {
            const uint4 x = *(const uint4*)(C.text + g);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) ascii |= (((w[k >> 2] >> (8 * (k & 3) + 7)) & 1u) ^ 1u) << k;
        }
        uint32_t stopm = (~(tk_member16(mk.p, TK_CB(TK_C_NU)) & ascii) | mk.hard) & 0xFFFFu;
        if (g + 16 <= p) stopm = 0;
        else if (g < p) stopm &= ~((2u << (uint32_t)(p - g)) - 1u);  // (p itself starts a piece: its own hard flag does not stop the run)
        else if (g == p) stopm &= ~1u;
        const uint32_t first = stopm ? (uint32_t)(g - wb) + (uint32_t)__ffs((int)stopm) - 1u : TKF_NONE;
        const uint32_t res = tk_coop_first(C, first, false);
        if (res != TKF_NONE) {
            r2 = wb + res;
            break;
        }
        if (wb + 4096 >= target + 4096) {  // far enough: the run reaches beyond the target
            r2 = wb + 4096;
            break;
        }
    }
    const uint64_t lim = r2 < target ? r2 : target;
    const uint64_t k = C.pat.digits();  // (the caller asks only for patterns with bounded digit groups)
    return lim > p ? p + k * ((lim - p) / k) : p;
}

// 8 bytes of the LDS text copy starting at byte offset o (three aligned dword reads)
__device__ __forceinline__ uint64_t tk_lds_load8(const uint8_t* raw, uint32_t o) {
    const uint32_t* dw = (const uint32_t*)raw;
    const uint32_t wi = o >> 2, sft = o & 3u;
    const uint32_t d0 = dw[wi], d1 = dw[wi + 1], d2 = dw[wi + 2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sft), hi = __builtin_amdgcn_alignbyte(d2, d1, sft);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t tk_key_of_lds(const uint8_t* raw, uint32_t o, uint32_t len) {
    if (len <= 8u) return tk_mask_low_bytes(tk_lds_load8(raw, o), len);
    if (len > TK_KEY_SAMPLED) {  // (tk_common.h: the length and four words)
        uint64_t h = TK_HASH_SEED ^ ((uint64_t)len << 32);
        h = tk_hash_step(h, tk_lds_load8(raw, o));
        h = tk_hash_step(h, tk_lds_load8(raw, o + 8u));
        h = tk_hash_step(h, tk_lds_load8(raw, o + len - 16u));
        h = tk_hash_step(h, tk_lds_load8(raw, o + len - 8u));
        return h == TK_EMPTY_KEY ? 0 : h;
    }
    uint64_t h = TK_HASH_SEED;
    uint32_t i = 0;
    for (; i + 8u <= len; i += 8u) h = tk_hash_step(h, tk_lds_load8(raw, o + i));
    if (i < len) h = tk_hash_step(h, tk_mask_low_bytes(tk_lds_load8(raw, o + i), len - i));
    if (h == TK_EMPTY_KEY) h = 0;
    return h;
}

// a piece for the tree kernel: reserve its scratch (4 uint32 per byte + the 64-ary min-tree levels)
__device__ __forceinline__ void tk_append_tree(uint32_t* listC, uint32_t* counters, uint32_t mi, uint32_t s, uint32_t len) {
    uint32_t lv = 0, c = len;
    do {
        c = (c + 63) >> 6;
        lv += c;
    } while (c > 64);
    const uint32_t gi = atomicAdd(&counters[TK_CNT_C], 1u);
    uint32_t* e = listC + 5 * (uint64_t)gi;
    e[0] = mi;
    e[1] = s;
    e[2] = len;
    e[3] = atomicAdd(&counters[TK_CNT_CBYTES], (len + 3u) & ~3u);  // (scratch offsets stay multiples of four entries: 16-byte loads)
    e[4] = atomicAdd(&counters[TK_CNT_CLEVELS], lv);
}

// window provider over the tile's LDS bitmaps: 64 positions from (segment wi, bit sh); kinds extracted on demand
struct TkWinLds {
    const uint64_t (*bm)[TK2_NSEG + 2];
    uint32_t wi, sh;
    uint64_t start, stop;
    __device__ __forceinline__ uint64_t get(int kind) const { return sh ? ((bm[kind][wi] >> sh) | (bm[kind][wi + 1] << (64u - sh))) : bm[kind][wi]; }
    __device__ __forceinline__ TkWinLds(const uint64_t (*bm_)[TK2_NSEG + 2], uint32_t wi_, uint32_t sh_) : bm(bm_), wi(wi_), sh(sh_) {
        start = get(TKB_START);
        stop = get(TKB_HARD) & ~1ull;
    }
};

struct TkWinLds32 {  // 32 positions from window offset r (the bitmaps seen as 32-bit words; one funnel = one v_alignbit)
    const uint32_t (*bm)[2 * (TK2_NSEG + 2)];
    uint32_t wi, sh;
    uint32_t start, stop;
    __device__ __forceinline__ uint32_t get(int kind) const { return __builtin_amdgcn_alignbit(bm[kind][wi + 1], bm[kind][wi], sh); }
    __device__ __forceinline__ TkWinLds32(const uint32_t (*bm_)[2 * (TK2_NSEG + 2)], uint32_t r) : bm(bm_), wi(r >> 5), sh(r & 31u) {
        start = get(TKB_START);
        stop = get(TKB_HARD) & ~1u;
    }
};

// eight text bytes at any position: one 12-byte load and two byte-aligns, no branch (the text is readable 64 bytes past its end)
__device__ __forceinline__ uint64_t tk_text_load8(const uint8_t* __restrict__ text, uint64_t pos) {
    const uintptr_t a = (uintptr_t)(text + pos);
    const uint32_t* w = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t sft = (uint32_t)a & 3u;
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    return ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sft) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sft);
}
// exact compare of the piece at LDS offset o with text[pos .. pos+len) in HBM: 8 * TKF_CMP_WORDS bytes per step, the loads of a step in flight
// together and no way out between them (round 4 compared eight bytes per step and left at the first difference: a chain of len / 8
// dependent loads in the one phase of the kernel that waits for memory; what lies behind the piece on either side is masked away)
#ifndef TKF_CMP_WORDS
#define TKF_CMP_WORDS 2
#endif
__device__ __forceinline__ bool tk_equal_lds_text(const uint8_t* raw, uint32_t o, const uint8_t* __restrict__ text, uint64_t pos, uint32_t len) {
    for (uint32_t i = 0; i < len; i += 8u * TKF_CMP_WORDS) {
        uint64_t g[TKF_CMP_WORDS];
#pragma unroll
        for (int j = 0; j < TKF_CMP_WORDS; ++j) g[j] = tk_text_load8(text, pos + i + 8u * (uint32_t)j);
        uint64_t diff = 0;
#pragma unroll
        for (int j = 0; j < TKF_CMP_WORDS; ++j) diff |= tk_keep_bytes(tk_lds_load8(raw, o + i + 8u * (uint32_t)j) ^ g[j], (int)len - (int)(i + 8u * (uint32_t)j));
        if (diff) return false;
    }
    return true;
}

// SLOW = false: one workgroup per tile.  A tile that needs the workgroup-wide scanner (tk_coop_*: no certain start in its left context, or
// a piece that leaves its window -- about 3 % of the tiles of ordinary text) is not finished here: it goes on the `deferred` list.
// SLOW = true: the same kernel with that scanner compiled in (its calls cost registers: kept out of the common path), launched over the
// deferred tiles with a fixed grid.
// Workgroups per CU.  20 KiB of LDS per tile x 8 = the CU's 160 KiB, and 8 x 4 wavefronts = its 32 wavefront slots: the kernel needs all
// of them (measured, profiles/r03_lds_cache_experiment.jsonl: a workgroup takes ~39 us per tile however many share the CU, so the rate
// is proportional to the workgroups in flight -- 5.9 ms per GiB with 8 per CU, 10.6 with 4, 13.6 with 3).  That is why there is no LDS-
// resident hot set of the vocabulary here (built and measured in rounds 3 and 4, removed in round 5: profiles/r04_lds_hot_set_closeout.txt --
// its 16-32 KiB per workgroup cost more in occupancy than its hits saved).
#ifndef TKF_OCC
#define TKF_OCC 8
#endif
#ifndef TKF_GIVEN_OCC
#define TKF_GIVEN_OCC TKF_OCC  // ... of the instance that finishes the deferred tiles (round 6: its loop over the list spills 26 registers at eight per CU; at six or five it spills fewer or none and is no faster -- 0.068 ms either way, tools/gpu_r6_call6.sh)
#endif
#ifndef TKF_SLOW_OCC
#define TKF_SLOW_OCC 3  // workgroups per CU of the deferred-tile variant (its grid: tk_api.hip, stage_deferred).  Round 6, on one box (tools/gpu_slowocc.sh):
                        // 3 against 4 -- C2 78.5 / 77.0 GB/s, C5 its kernel 0.168 / 0.188 ms, C3 0.40 / 0.38 ms (within the noise of the box)
#endif
// A pointer put together from an integer (the kernarg segment read again, a word parked in LDS) is a GENERIC pointer to the compiler: its loads and stores
// become flat_* instructions -- a 64-bit address in vector registers for every access (no scalar base + 32-bit offset form), and counted by lgkmcnt as well as
// vmcnt, so that every wait for an LDS read also waits for the table probes in flight.  Through a pointer of the global address space they are global_* again.
#ifndef TKF_GLOBAL_PTRS
#define TKF_GLOBAL_PTRS 1
#endif
#if TKF_GLOBAL_PTRS
#define TKF_PTR(type, v) ((type)(__attribute__((address_space(1))) void*)(uintptr_t)(v))
#else
#define TKF_PTR(type, v) ((type)(uintptr_t)(v))
#endif
#ifndef TKF_CLAIM_SPIN
#define TKF_CLAIM_SPIN 8  // looks a duplicate takes at a slot whose claimant has not written its words yet (see `claim`)
#endif
// Experiments only (-DTKF_TIMING, tools/build_variant.sh): where a workgroup's time per tile goes.  Thread 0 reads the shader clock at the phase
// boundaries of the one-tile-per-workgroup instance and adds the differences up in tk_time_acc (read and reset through tk_stat "time_<i>" /
// "time_reset"); slot 15 counts the tiles.
#ifdef TKF_TIMING
__device__ unsigned long long tk_time_acc[2 * 1024 * 16];  // (spread over 1024 lines by workgroup: same-address atomics would be what is measured)
#define TKT(i)                                                                  \
    do {                                                                        \
        if ((MODE == TKF_MODE_TILE || MODE == TKF_MODE_STARTS) && tid == 0) {   \
            const unsigned long long tkt_now = __builtin_readcyclecounter();    \
            atomicAdd(&tk_time_acc[(MODE == TKF_MODE_STARTS ? 16384u : 0u) + (blockIdx.x & 1023u) * 16u + i], tkt_now - tkt_prev); \
            tkt_prev = tkt_now;                                                 \
        }                                                                       \
    } while (0)
#else
#define TKT(i) do { } while (0)
#endif
// (round 6, TKF_EXTEND) Where does the run of letters that reaches the end of a front-kernel window end?  One wavefront reads on from `from` (8-byte
// aligned), eight bytes per lane and 512 per step, a class lookup per char, up to 2 KiB.  CL (cl100k): letters of any kind, ended by anything else;
// otherwise (o200k) letters without case and marks, ended by anything but a cased letter or an apostrophe.  Returns the end relative to `from - TK2_WIN`
// (the window's base), or 0 when the run goes on differently or for too long.  The table and bitmap pointers come from LDS (ext_sh in tk_k_front): as
// kernel arguments kept alive up to this rare path they cost the kernel 2 % (4.38 -> 4.46 ms per GiB: 135 more reloads of spilled scalar registers in the
// paths that every tile takes); a call instead of inlined code changed nothing.
template <bool CL>
__device__ __forceinline__ uint32_t tk_extend_letter_run(const uint8_t* __restrict__ uc_bmp, const uint8_t* __restrict__ uc_stage1, const uint8_t* __restrict__ uc_stage2, const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ brk, uint64_t from, int lane, bool seen_ll) {
    // (o200k: the run may go on with lower-case letters, LL.  From the first of them on the matcher is in the alternative's lower-case part, where an upper-case
    // letter ENDS the piece; before it -- LC and MK are in both parts, the state is not known -- an upper-case letter lets the piece go on: not decided here.
    // seen_ll: the window's last char was one.)
    constexpr uint32_t SET = CL ? TK_M_L : (TK_CB(TK_C_LC) | TK_CB(TK_C_MK) | TK_CB(TK_C_LL));
    auto cls_cp = [&](uint32_t cpt) -> uint32_t {
        uint32_t cl = uc_bmp[cpt < 0x10000u ? cpt : 0xFFFFu];
        if (cpt >= 0x10000u && cpt <= 0x10FFFFu) cl = uc_stage2[(uint32_t)uc_stage1[cpt >> 8] * 256u + (cpt & 255u)];
        return cl;
    };
    for (uint32_t it = 0; it < 4u; ++it) {
        const uint64_t g = from + 512u * it + 8u * (uint32_t)lane;
        uint32_t d0 = 0, d1 = 0, d2 = 0, hb = 0;
        if (g < n) {  // (the text is readable 64 bytes past its end)
            const uint2 v = *(const uint2*)(text + g);
            d0 = v.x;
            d1 = v.y;
            d2 = *(const uint32_t*)(text + g + 8);
            hb = (brk[g >> 5] >> (g & 31u)) & 0xFFu;  // hard starts (documents, special tokens) among the eight positions
        }
        uint32_t bad = 8u, badc = 0u, llm = 0u;  // (llm: the lane's char starts of class LL)
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            const uint32_t four = j == 0 ? d0 : (j < 4 ? __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)j) : (j == 4 ? d1 : __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)(j - 4))));
            const uint32_t b0 = four & 0xFFu;
            uint32_t ln;
            const uint32_t cpt = b0 < 0x80u ? b0 : tk_utf8_cp(four, &ln);
            uint32_t cl = cls_cp((b0 & 0xC0u) == 0x80u ? 0u : cpt);
            if (g + (uint32_t)j >= n || ((hb >> j) & 1u)) cl = TK_C_END;
            if ((b0 & 0xC0u) != 0x80u || g + (uint32_t)j >= n) {  // a char start (or the end of the text)
                if (!((SET >> cl) & 1u)) {
                    bad = (uint32_t)j;
                    badc = cl;
                }
                if (!CL && cl == (uint32_t)TK_C_LL) llm |= 1u << j;
            }
        }
        const uint64_t m = __ballot(bad < 8u);
        if (m) {
            const int l2 = __ffsll((unsigned long long)m) - 1;
            const uint32_t bj = (uint32_t)__shfl((int)bad, l2, 64), bc = (uint32_t)__shfl((int)badc, l2, 64);
            if constexpr (!CL) {
                if (bc == (uint32_t)TK_C_AP) return 0u;  // (a contraction may follow)
                if (bc == (uint32_t)TK_C_LU) {
                    const bool before = lane < l2 ? llm != 0u : (lane == l2 && (llm & ((1u << bj) - 1u)) != 0u);
                    if (!seen_ll && __ballot(before) == 0ull) return 0u;
                }
            }
            return (uint32_t)TK2_WIN + 512u * it + 8u * (uint32_t)l2 + bj;
        }
        if (!CL && __ballot(llm != 0u) != 0ull) seen_ll = true;
    }
    return 0u;
}
// The kernel's arguments as the kernarg segment lays them out (TKF_PARK_ARGS == 2: phase E reads what it needs of them from there again, by scalar loads
// at the point of use, instead of carrying them in registers from the kernel's entry; the static_asserts in tk_k_front tie the offsets to the signature).
struct TkFrontArgs {
    TkTables T;
    const uint8_t* text;
    uint64_t n, chunk_base;
    const uint32_t *brk, *docb, *ss, *si;
    TkFrontOut out;
    TkMissKey* mt;
    uint32_t mt_mask;
    uint32_t* deferred;
    const uint32_t* gapb;
    int dbg;
};
template <int PAT, bool SPEC, int MODE>
__global__ __launch_bounds__(256, MODE == TKF_MODE_STARTS ? TKF_SLOW_OCC : (MODE == TKF_MODE_GIVEN ? TKF_GIVEN_OCC : TKF_OCC)) void tk_k_front(TkTables T, const uint8_t* __restrict__ text, uint64_t n, uint64_t chunk_base,
                                                  const uint32_t* __restrict__ brk, const uint32_t* __restrict__ docb,
                                                  const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si, TkFrontOut out,
                                                  TkMissKey* __restrict__ mt, uint32_t mt_mask, uint32_t* __restrict__ deferred,
                                                  const uint32_t* __restrict__ gapb /* gap chars of the generic engine's split, or null */, int dbg) {
    // the pattern: a compile-time constant for the three stock patterns; PAT = TK_PAT_GENERIC reads family and parameters from the tables
    constexpr bool GEN = PAT == TK_PAT_GENERIC;
    constexpr bool SLOW = MODE == TKF_MODE_STARTS, GIVEN = MODE == TKF_MODE_GIVEN;
    const TkPat pat = GEN ? T.pat : tk_stock_pat(PAT);
    const int fam = GEN ? T.pat.fam() : PAT;
    constexpr int NW = TK2_NSEG + 2;            // 64-bit words per bitmap (two sentinel words beyond the window)
    constexpr int BM_BYTES = TKB_KINDS * NW * 8;
    __shared__ __attribute__((aligned(16))) uint8_t raw[TK2_WIN + 16];
    __shared__ __attribute__((aligned(16))) uint8_t pool[BM_BYTES + TK2_CLIST * 2];  // bitmaps + certain list; later the piece list
    __shared__ __attribute__((aligned(8))) uint64_t planes[4][NW];                   // class planes (bit = text byte)
    __shared__ __attribute__((aligned(8))) uint32_t btab[256 * 2];                   // byte table (tk_chunk.h)
    __shared__ uint32_t certw[TK2_WIN / 32];                                         // certain starts (hard starts included)
    __shared__ uint32_t bits[TK_TILE / 32];
    __shared__ uint8_t lastc_own[256];
    __shared__ uint32_t np_sh, need_walk, last_end_sh, ncont_sh, ngap_sh;
    __shared__ __attribute__((aligned(8))) uint16_t contl_own[SLOW ? TKF_CONT_CAP : 4], stop_own[SLOW ? 256 : 4];  // (read as 32-bit words)
    // second window of the deferred-tile variant: a stretch of text left of the tile, walked by tk_coop_window_walk
    __shared__ __attribute__((aligned(16))) uint8_t w2_raw[SLOW ? TK2_WIN + 16 : 16];
    __shared__ uint32_t w2_planes[SLOW ? 4 * TK2_PLW : 1], w2_start[SLOW ? TK2_WIN / 32 + 4 : 1], w2_hard[SLOW ? TK2_WIN / 32 + 4 : 1];
    __shared__ uint16_t w2_jump[SLOW ? TK2_WIN : 2];
    __shared__ uint16_t slowl[TKF_SLOW_CAP];  // pieces that leave the window (window positions of their starts)
    __shared__ uint32_t nslow_sh;
    // (the special-token bitmaps have no copy of their own -- two more arrays of this size took the kernel over an eighth of the CU's LDS:
    // phase C reads each lane's 16 bits from global memory, phase F finds the starts of special tokens in `brkw`, reloaded in phase E)
    __shared__ uint32_t brkw[TK2_WIN / 32 + 1];
    __shared__ uint32_t scan_sh[8];
    // (TKF_EXTEND) four pointers the rare way out of "a piece leaves the window" needs, parked in LDS by phase A: kept in scalar registers up to that point
    // of the kernel they made it spill 135 more scalar reloads into the paths that every tile takes (4.38 -> 4.46 ms per GiB)
    __shared__ uint64_t ext_sh[4];
#ifndef TKF_PARK_ARGS
#define TKF_PARK_ARGS 2  // 0: the arguments stay in registers; 1: parked in LDS (park_sh); 2: read from the kernarg segment again where phase E begins
#endif
    // (experiment, round 6) what only phases E and F need of the kernel's arguments, parked in LDS by phase A and read back where phase E begins: kept in scalar
    // registers from the kernel's first instruction they are part of the 150 scalar values the kernel spills into vector-register lanes
    // ... and better still (TKF_PARK_ARGS == 2, shipped): they are in memory already -- the kernarg segment -- and a scalar load at the point of use costs
    // neither LDS traffic nor the thirty-four v_readfirstlane of the LDS form: front 4.39 -> 4.35 ms per GiB on one box.  The compiler must not see that the
    // pointer is the kernarg segment's (it would load at the kernel's entry again): it goes through an empty asm statement.  tk_k_front_args_check
    // (run once by tk_create) compares what the offsets of TkFrontArgs give with the arguments themselves.
#if TKF_PARK_ARGS == 1
    __shared__ uint64_t park_sh[18];
#endif
    uint64_t(*bm)[NW] = (uint64_t(*)[NW])pool;
    uint8_t* lastc = lastc_own;
    // From phase C on the byte table is dead in the one-tile-per-workgroup variant: its 2 KiB hold the "stop" bitmap of the scanners'
    // short cut (256 halfwords) and the continuation list; the deferred-tile variant still classifies text with it (tk_coop_chunk).
    constexpr uint32_t CONT_CAP = SLOW ? (uint32_t)TKF_CONT_CAP : (uint32_t)TKF_CONT_CAP_FAST;
    uint16_t* stop16 = SLOW ? stop_own : (uint16_t*)btab;          // [256] char starts at which a piece may start (16 per lane)
    uint16_t* contl = SLOW ? contl_own : (uint16_t*)btab + 256;    // [CONT_CAP] scan chains still to be walked (window positions)
    const uint32_t tid = threadIdx.x;
#ifndef TKF_SCALAR_WID
#define TKF_SCALAR_WID 1
#endif
    const int lane = tid & 63, wid = TKF_SCALAR_WID ? __builtin_amdgcn_readfirstlane((int)(tid >> 6)) : (int)(tid >> 6);
    uint32_t item = blockIdx.x;
#ifdef TKF_TIMING
    unsigned long long tkt_prev = __builtin_readcyclecounter();
    if (MODE == TKF_MODE_TILE && tid == 0) atomicAdd(&tk_time_acc[(blockIdx.x & 1023u) * 16u + 15u], 1ull);
#endif
    // SLOW: persistent, a fixed grid walks the deferred list with the stride of the grid.  Otherwise one tile per workgroup (the loop's
    // state would cost registers the kernel does not have at eight workgroups per CU).
    // (round 6) GIVEN as well: the host launches it without having read the list's length (a grid from the chunk before, stage_deferred) and the grid
    // walks the list with its stride -- an instance of its own, whose loop state costs the common instance nothing.
    constexpr bool PERSIST = SLOW || GIVEN;
    const uint32_t n_items = SLOW ? out.counters[(dbg & TKF_DBG_SECOND) ? TK_CNT_DEFER2 : TK_CNT_DEFER] : (GIVEN ? (uint32_t)__builtin_amdgcn_readfirstlane((int)out.counters[TK_CNT_DEFER]) : gridDim.x);
    if (PERSIST && item >= n_items) return;
    // (round 6) The deferred tiles differ a lot -- a walk back through a megabyte of one class, or nothing of the kind: 30 000 to 400 000 cycles -- and
    // there are few of them (0.75 % of the tiles of web text: 2.7 per workgroup of this grid).  With the stride of the grid the kernel lasted as long as
    // the workgroup with the three dearest; now a workgroup takes its first tile by its index and every further one from a counter.
    __shared__ uint32_t next_item_sh;
    auto next_item = [&]() -> bool {
        if constexpr (GIVEN) {
            item += gridDim.x;
            return item < n_items;
        }
        __syncthreads();
        if (tid == 0) next_item_sh = gridDim.x + atomicAdd(&out.counters[TK_CNT_SLOWQ + ((dbg & TKF_DBG_SECOND) ? 1 : 0)], 1u);
        __syncthreads();
        item = next_item_sh;
        return item < n_items;
    };
    do {
    if (PERSIST && item != blockIdx.x) __syncthreads();  // (the shared arrays are reused by the next tile)
#ifdef TKF_TIMING
    if (MODE == TKF_MODE_STARTS && tid == 0) {
        atomicAdd(&tk_time_acc[16384u + (blockIdx.x & 1023u) * 16u + 15u], 1ull);
        tkt_prev = __builtin_readcyclecounter();
    }
#endif
    const uint64_t tile = (SLOW || GIVEN) ? (uint64_t)deferred[item] : (uint64_t)item;
    auto defer_tile = [&]() {
        if (tid == 0) deferred[atomicAdd(&out.counters[TK_CNT_DEFER], 1u)] = (uint32_t)tile;
#ifdef TKF_TIMING
        if (tid == 0) atomicAdd(&tk_time_acc[(blockIdx.x & 1023u) * 16u + 14u], need_walk ? 1ull : (1ull << 32));  // (experiments: why tiles are deferred -- low word: no certain start in the left context; high word: a piece leaves the window)
#endif
    };
    const uint64_t tile_start = tile * TK_TILE;
    const uint64_t tile_end = tile_start + TK_TILE < n ? tile_start + TK_TILE : n;
    const int64_t base = (int64_t)tile_start - TK2_LEFT;
#ifndef TKF_FRESH_SCALARS
#define TKF_FRESH_SCALARS 0  // experiment (see below): 0 off, 1 in phases E and F, 2 in the whole kernel
#endif
#if TKF_FRESH_SCALARS == 2
    // (round 6) What the compiler derives from the tile's index and keeps for the whole kernel -- the tile's first byte (tile x TK_TILE, 64 bits), its end, the
    // window's base -- were scalar registers spilled at the kernel's entry and reloaded by vector instructions wherever they are used (74 of the kernel's 150
    // v_readlane).  Through an empty asm statement at the point of use they are two scalar multiplications there and nothing is kept but the index.
    auto tile_start_fresh = [&]() -> uint64_t { uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tile); asm volatile("" : "+s"(t)); return (uint64_t)t * (uint64_t)TK_TILE; };
    auto tile_end_fresh = [&]() -> uint64_t { const uint64_t e = tile_start_fresh() + TK_TILE; return e < n ? e : n; };
    auto base_fresh = [&]() -> int64_t { return (int64_t)tile_start_fresh() - TK2_LEFT; };
#define tile_start tile_start_fresh()
#define tile_end tile_end_fresh()
#define base base_fresh()
#endif
    // ---- A: every lane owns 16 bytes of the window (kept in registers and copied to LDS)
    const int64_t gp = base + (int64_t)tid * 16;
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t valid = 0, past = 0;  // 16-bit masks: byte is text / byte lies at or after the end of the text
    if (gp + 16 > 0 && (uint64_t)(gp < 0 ? 0 : gp) < n) {
        const uint4 x = *(const uint4*)(text + gp);  // (a window never starts before -64: gp >= 0 here except on tile 0, where gp + 16 > 0 => gp >= 0)
        w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
        const uint64_t left = n - (uint64_t)gp;       // text bytes from gp on
        valid = left >= 16 ? 0xFFFFu : ((1u << (uint32_t)left) - 1u);
        if (left < 16) {                              // the text ends inside the chunk: bytes after it read as zero
            const uint32_t nb = (uint32_t)left;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int lo = 4 * d;
                const uint32_t keep = nb >= (uint32_t)lo + 4u ? 0xFFFFFFFFu : (nb <= (uint32_t)lo ? 0u : ((1u << (8u * (nb - lo))) - 1u));
                w[d] &= keep;
            }
        }
    }
    if (gp >= 0) past = ~valid & 0xFFFFu;  // (positions before the text, tile 0 only, are neither text nor "past")
    *(uint4*)(raw + tid * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    if (tid == 0) {  // 16 more bytes behind the window: a char that starts in its last bytes is decoded whole
        const int64_t ge = base + TK2_WIN;
        uint4 x = make_uint4(0, 0, 0, 0);
        if ((uint64_t)ge < n) x = *(const uint4*)(text + ge);  // (bytes past n only follow a complete char: never decoded)
        *(uint4*)(raw + TK2_WIN) = x;
    }
    *(uint2*)&btab[tid * 2] = *(const uint2*)&T.byte_tab[tid * 2];
    if (tid < TK2_WIN / 32) {  // break / special bitmaps of the window (the window base is 32-aligned)
        int64_t wgp = base + (int64_t)tid * 32;
        bool in = wgp >= 0 && (uint64_t)wgp < n;
        brkw[tid] = in ? brk[wgp >> 5] : 0u;
    }
    if (tid == 0) {
        if (!SLOW && !GIVEN) {
            ext_sh[0] = (uint64_t)(uintptr_t)brk;
            ext_sh[1] = (uint64_t)(uintptr_t)T.uc_bmp;
            ext_sh[2] = (uint64_t)(uintptr_t)T.uc_stage1;
            ext_sh[3] = (uint64_t)(uintptr_t)T.uc_stage2;
        }
#if TKF_PARK_ARGS == 1
        if (!SLOW) {
            park_sh[0] = (uint64_t)(uintptr_t)T.short_tab;
            park_sh[1] = (uint64_t)(uintptr_t)T.mid_tab;
            park_sh[2] = (uint64_t)(uintptr_t)T.xl;
            park_sh[3] = (uint64_t)(uintptr_t)T.tok_bytes;
            park_sh[4] = (uint64_t)(uintptr_t)mt;
            park_sh[5] = (uint64_t)(uintptr_t)out.res;
            park_sh[6] = (uint64_t)(uintptr_t)out.data.tab;
            park_sh[7] = (uint64_t)(uintptr_t)out.data.ovf;
            park_sh[8] = (uint64_t)(uintptr_t)out.listC;
            park_sh[9] = (uint64_t)T.short_mask | ((uint64_t)T.short_shift << 32);
            park_sh[10] = (uint64_t)T.mid_mask | ((uint64_t)T.mid_shift << 32);
            park_sh[11] = (uint64_t)T.xl_mask | ((uint64_t)T.max_token_len << 32);
            park_sh[12] = (uint64_t)mt_mask | ((uint64_t)out.data.ovf_base << 32);
            park_sh[13] = (uint64_t)out.ovf_cap;
            park_sh[14] = (uint64_t)(uintptr_t)T.piece;  // (the probe of a piece of more than TK_XL_MAX bytes where there is no in-call table: small chunks)
            park_sh[15] = (uint64_t)(uintptr_t)T.piece_off;
            park_sh[16] = T.piece_mask;
        }
#endif
        need_walk = 0;
        ncont_sh = 0;
        nslow_sh = 0;
        last_end_sh = (uint32_t)(tile_end - tile_start) + TK2_LEFT;
    }
    if (tid < TKB_KINDS) {
        bm[tid][TK2_NSEG] = tid <= TKB_HARD ? ~0ull : 0ull;  // beyond the window: unknown -> "stop"
        bm[tid][TK2_NSEG + 1] = tid <= TKB_HARD ? ~0ull : 0ull;
    }
    if (tid < 4) planes[tid][TK2_NSEG] = planes[tid][TK2_NSEG + 1] = tid >= 2 ? ~0ull : 0ull;  // (class END)
    __syncthreads();
    TKT(0);
    if (dbg & 0x1000) {  // (perf experiments: stop after this phase)
        if (tid == 0) out.tile_np[tile] = 0;
        continue;
    }
#ifndef TKF_HARD_ONLY
#define TKF_HARD_ONLY 1
#endif
    if constexpr (GIVEN) {
        // ---- the tile's piece starts and the end of its last piece are given (the deferred-tile instance has found them)
        if (tid < TK_TILE / 32) {
            const uint64_t wgp = tile_start / 32 + tid;
            bits[tid] = wgp * 32 < n ? out.starts[wgp] : 0u;
        }
        if (tid == 0) last_end_sh = out.tile_np[tile];
        __syncthreads();
    } else if (TKF_HARD_ONLY && !SLOW && (dbg & TKF_DBG_HARD_ONLY)) {
        // ---- hard starts only: the pieces that start in the tile begin at its hard starts; the last of them ends at the first hard start
        // at or behind the tile's end -- in the 128 bytes of look-ahead, or the tile is one for the workgroup-wide scanner
        const uint32_t te = (uint32_t)(tile_end - tile_start) + (uint32_t)TK2_LEFT;
        uint32_t mine_w = 0;
        if (tid < TK_TILE / 32) {
            const uint32_t lo = (uint32_t)TK2_LEFT + tid * 32u;  // window position of the word's first bit
            mine_w = brkw[lo >> 5];
            if (lo + 32u > te) mine_w &= lo >= te ? 0u : ((1u << (te - lo)) - 1u);
            bits[tid] = mine_w;
            if (mine_w) need_walk = 1u;  // (the flag of the other branch, zero since phase A: here "the tile has a piece start"; no __syncthreads_or: it costs 256 B of LDS)
        }
        __syncthreads();
        if (need_walk && tile_end < n) {  // (a tile that ends with the text keeps last_end_sh = the end of the text)
            if (tid == 0) {
                uint32_t e = 0xFFFFFFFFu;
                for (uint32_t w = te >> 5; w < (uint32_t)TK2_WIN / 32u && e == 0xFFFFFFFFu; ++w) {  // (te is a multiple of 32 here: a full tile)
                    const uint32_t v = brkw[w];
                    if (v) e = w * 32u + (uint32_t)__ffs((int)v) - 1u;
                }
                last_end_sh = e;
            }
            __syncthreads();
            if (last_end_sh == 0xFFFFFFFFu) {  // the last piece leaves the window
                // (round 6) ... and ends at the next hard start all the same: one wavefront reads the break bitmap on, 2048 positions a step, up to 8 KiB
                // (the tiles of the generic engine's split that went to the workgroup-wide scanner for this: 0.32 of its 11.0 ms per GiB)
                if (wid == 0) {
                    const uint32_t* brk_g = TKF_PTR(const uint32_t*, ext_sh[0]);
                    const uint64_t w0 = (uint64_t)(base + TK2_WIN) >> 5, nw = (n + 31) >> 5;
                    uint32_t e = 0xFFFFFFFFu;
                    for (uint32_t it = 0; it < 4u && e == 0xFFFFFFFFu; ++it) {
                        const uint64_t wq = w0 + 64u * it + (uint32_t)lane;
                        const uint32_t v = wq < nw ? brk_g[wq] : 0u;
                        const uint64_t m = __ballot(v != 0u);
                        if (m) {
                            const int l2 = __ffsll((unsigned long long)m) - 1;
                            e = (uint32_t)TK2_WIN + (64u * it + (uint32_t)l2) * 32u + (uint32_t)__ffs((int)(uint32_t)__shfl((int)v, l2, 64)) - 1u;
                        } else if (w0 + 64u * it + 64u >= nw) {
                            e = (uint32_t)(n - (uint64_t)base);  // (no hard start up to the end of the text: the piece ends with it)
                        }
                    }
                    if (lane == 0) last_end_sh = e;
                }
                __syncthreads();
                if (last_end_sh == 0xFFFFFFFFu) {
                    defer_tile();
                    continue;
                }
            }
        }
    } else {
    // ---- B: classes of the lane's 16 bytes as 16-bit masks (tk_chunk.h): table pass, decode pass for non-ASCII chars
    TkChunk ch;
    {
        auto tab = [&](uint32_t b, uint32_t& x, uint32_t& y) {
            const uint2 e = *(const uint2*)&btab[b * 2];
            x = e.x;
            y = e.y;
        };
        tk_chunk_table_pass(w, tab, ch);
        const uint32_t* dw = (const uint32_t*)raw;
        auto get4 = [&](int k) -> uint32_t {  // four window bytes at chunk-relative offset k (tid > 0 whenever k < 0)
            const uint32_t o = tid * 16u + (uint32_t)k;
            return __builtin_amdgcn_alignbyte(dw[(o >> 2) + 1], dw[o >> 2], o & 3u);
        };
        auto cls_of = [&](uint32_t cp) -> uint32_t {  // (one load below U+10000 -- no `if` around it: the loads of a step are in flight together)
            uint32_t cl = T.uc_bmp[cp < 0x10000u ? cp : 0xFFFFu];
            if (cp >= 0x10000u && cp <= 0x10FFFFu) cl = T.uc_stage2[(uint32_t)T.uc_stage1[cp >> 8] * 256u + (cp & 255u)];
            return cl;
        };
        const uint32_t prev = tid ? dw[tid * 4u - 1u] : 0u;
#ifndef TKF_DENSE_DECODE
#define TKF_DENSE_DECODE 1
#endif
        bool decoded = false;
        if constexpr (TKF_DENSE_DECODE && MODE == TKF_MODE_TILE) {
            // (round 6) The non-ASCII chars of a wavefront's 1024 bytes by DENSE lanes.  tk_chunk_decode is a loop of ~100 vector instructions per two chars of
            // a lane that runs as long as the lane with the most chars (up to eight in sixteen bytes) while the average lane of web text has fewer than two:
            // most of its instructions are issued for a handful of active lanes.  Here the lanes list the window positions of their lead bytes (a prefix sum
            // of the counts; five instructions per own lead), lane i decodes the i-th listed char whichever lane holds it and ORs its class into the
            // wavefront's span of the class planes (LDS atomics on words nobody else touches before phase C rewrites them from the registers), and every lane
            // reads its sixteen bits of the four planes back.  A char that straddles two lanes is decoded once (by the lead's position); the char that
            // straddles in from the wavefront before is an entry of the wavefront's first lane, as in tk_chunk_decode.
            const uint32_t cont = tk_plane16(ch.f0, ch.f1, 0);
            const uint32_t leads16 = tk_plane16(ch.f0, ch.f1, 1) | tk_plane16(ch.f0, ch.f1, 2) | tk_plane16(ch.f0, ch.f1, 3);
            const bool pend0 = (cont & 1u) && tid > 0 && lane == 0;
            decoded = true;
            if (__ballot(leads16 != 0u || pend0)) {
                const uint32_t cnt = (uint32_t)__popc(leads16) + (pend0 ? 1u : 0u);
                const uint32_t inc = tk_wave_scan_u32(cnt, lane);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
#ifndef TKF_DENSE_MAX
#define TKF_DENSE_MAX 128  // (at most 512: the list's size.  One box, front kernel in ms per GiB: 64 -> 4.27, 96 -> 4.29, 128 -> 4.24, 192 -> 4.26, 512 -> 4.33; without: 4.28-4.29)
#endif
                if (total > (uint32_t)TKF_DENSE_MAX) {
                    decoded = false;  // (text that is all non-ASCII: every lane is busy in tk_chunk_decode's loop as well, and its step costs less per char)
                } else {
                    uint16_t* dl = (uint16_t*)(pool + BM_BYTES) + (uint32_t)wid * 512u;  // (the certain list's place: unused before phase E)
                    uint32_t* pl32 = (uint32_t*)planes;                                   // (rows of 2 * NW words; words 32 wid .. 32 wid + 31 are this wavefront's)
                    constexpr uint32_t ROW = 2u * (uint32_t)NW;
                    const uint32_t w0 = 32u * (uint32_t)wid;
                    pl32[((uint32_t)lane >> 5) * ROW + w0 + ((uint32_t)lane & 31u)] = 0u;
                    pl32[(((uint32_t)lane >> 5) + 2u) * ROW + w0 + ((uint32_t)lane & 31u)] = 0u;
                    uint32_t o = inc - cnt;
                    if (pend0) {
                        const int kprev = (prev >> 24) >= 0xC0u ? -1 : (((prev >> 16) & 0xFFu) >= 0xC0u ? -2 : -3);
                        dl[o++] = (uint16_t)(tid * 16u + (uint32_t)kprev);
                    }
                    for (uint32_t m = leads16; m; m &= m - 1u) dl[o++] = (uint16_t)(tid * 16u + (uint32_t)__ffs((int)m) - 1u);
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
                        const uint32_t e = e0 + (uint32_t)lane;
                        if (e < total) {
                            const uint32_t pos = dl[e];
                            uint32_t len;
                            const uint32_t cp = tk_utf8_cp(__builtin_amdgcn_alignbyte(dw[(pos >> 2) + 1], dw[pos >> 2], pos & 3u), &len);
                            const uint32_t cl = cls_of(cp);
                            int rel = (int)pos - (int)(1024u * (uint32_t)wid);
                            uint32_t ones = (1u << len) - 1u;
                            if (rel < 0) {
                                ones >>= (uint32_t)(-rel);
                                rel = 0;
                            }
                            const uint32_t wq = (uint32_t)rel >> 5;
                            const uint64_t M = (uint64_t)ones << ((uint32_t)rel & 31u);
                            const uint32_t lo = (uint32_t)M, hi = wq < 31u ? (uint32_t)(M >> 32) : 0u;
#pragma unroll
                            for (uint32_t pq = 0; pq < 4u; ++pq) {
                                const uint32_t bit = 0u - ((cl >> pq) & 1u);
                                atomicOr(&pl32[pq * ROW + w0 + wq], lo & bit);
                                if (hi & bit) atomicOr(&pl32[pq * ROW + w0 + wq + 1u], hi & bit);  // (a char across two words of the planes; all four behind ONE test of `hi`: slower)
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t hw16 = (tid & 1u) * 16u;
                    uint32_t v[4];
#pragma unroll
                    for (uint32_t pq = 0; pq < 4u; ++pq) v[pq] = (pl32[pq * ROW + (tid >> 1)] >> hw16) & 0xFFFFu;
                    ch.acc0 |= (v[0] & 0xFFu) | ((v[1] & 0xFFu) << 8) | ((v[2] & 0xFFu) << 16) | ((v[3] & 0xFFu) << 24);
                    ch.acc1 |= (v[0] >> 8) | ((v[1] >> 8) << 8) | ((v[2] >> 8) << 16) | ((v[3] >> 8) << 24);
                }
            }
        }
        if (!decoded) tk_chunk_decode(ch, prev, tid > 0, get4, cls_of);
    }
    TKT(1);
    if (dbg & 0x2000) {  // (perf experiments: stop after the classification)
        if (tid == 0 || (ch.acc0 ^ ch.acc1) == 0xFFFFFFF1u) out.tile_np[tile] = 0;
        continue;
    }
    // ---- C: masks of the chunk -> bitmaps in LDS; certain starts; the list of scan starts
    TkChunkMasks mk;
    {
        const uint32_t sh16 = (tid & 1u) * 16u;
        const uint32_t brk16 = (brkw[tid >> 1] >> sh16) & 0xFFFFu;
        uint32_t ss16 = 0u, si16 = 0u;
        if constexpr (SPEC) {  // (the window base is 32-aligned: the lane's 16 positions are one half of a bitmap word)
            const int64_t wgp = base + (int64_t)(tid >> 1) * 32;
            if (wgp >= 0 && (uint64_t)wgp < n) {
                ss16 = (ss[wgp >> 5] >> sh16) & 0xFFFFu;
                si16 = (si[wgp >> 5] >> sh16) & 0xFFFFu;
            }
        }
        tk_chunk_finalize(ch, valid, past, brk16, ss16, si16, mk);
    }
    TkSets st;
    tk_sets_from_planes(mk.p[0], mk.p[1], mk.p[2], mk.p[3], st);
    {
        const bool O2 = fam == TK_PAT_O200K, R5 = fam == TK_PAT_R50K;  // (constants unless GEN)
        uint16_t* b16 = (uint16_t*)pool;  // halfword tid of bitmap `kind` = positions [16 tid, 16 tid + 16)
        constexpr int HW = NW * 4;         // halfwords per bitmap
        b16[TKB_START * HW + tid] = (uint16_t)mk.start;
        b16[TKB_HARD * HW + tid] = (uint16_t)mk.hard;
        b16[TKB_OTH * HW + tid] = (uint16_t)st.oth;
        b16[TKB_WS * HW + tid] = (uint16_t)st.ws;
        b16[TKB_NU * HW + tid] = (uint16_t)st.nu;
        if (!O2 || GEN) b16[TKB_L * HW + tid] = (uint16_t)st.l;
        if (O2 || GEN) {
            b16[TKB_UP * HW + tid] = (uint16_t)st.up;
            b16[TKB_LOW * HW + tid] = (uint16_t)st.low;
            b16[TKB_CAS * HW + tid] = (uint16_t)st.cas;
            // (generic patterns: the suffix set behind a run of "other" chars, whatever the family)
            b16[TKB_NLSL * HW + tid] = (uint16_t)(GEN ? (((pat.suffix() & 1u) ? st.nl : 0u) | ((pat.suffix() & 2u) ? st.sl : 0u)) : st.nlsl);
        }
        if (!R5 || GEN) b16[TKB_NL * HW + tid] = (uint16_t)st.nl;
        uint16_t* p16 = (uint16_t*)planes;
#pragma unroll
        for (int p = 0; p < 4; ++p) p16[p * HW + tid] = (uint16_t)mk.p[p];
        lastc[tid] = (uint8_t)tk_class_from_planes(mk.p, 15);
    }
    __syncthreads();
    // the class before the chunk: never known for the first chunk of the window (nothing there is certain unless hard)
    const uint32_t prevc = tid ? (uint32_t)lastc[tid - 1] : 0u;
    uint32_t near = 0xFFFFu;  // positions with an apostrophe two or three bytes before (the first chunk of the window cannot know: all)
    {
        uint32_t apb = 7u;
        if (tid) {
            const uint32_t pv = ((const uint32_t*)raw)[tid * 4u - 1u];
            apb = (uint32_t)(((pv >> 8) & 0xFFu) == 0x27u) | ((uint32_t)(((pv >> 16) & 0xFFu) == 0x27u) << 1) | ((uint32_t)((pv >> 24) == 0x27u) << 2);
        }
        near = tk_chunk_near(st.ap, apb);
    }
    const uint32_t cert = GEN ? tk_chunk_certain_rt(T.cert, st, mk.text, mk.hard & mk.text, prevc) : tk_chunk_certain(fam, st, mk.text, mk.hard & mk.text, prevc, near);
    ((uint16_t*)certw)[tid] = (uint16_t)cert;
    // The scanners' short cut: between a piece start and the next position at which a piece MAY start there is no boundary.  "May
    // start" = every char start except those whose class pair never has one (tk_chunk_never; generic patterns: every char start).
    // (round 6) ... and the stops behind which a piece certainly goes on to the NEXT stop (tk_chunk.h, tk_chunk_second_stop): a bitmap of its
    // own, in the place of one the pattern's scanner does not use (o200k: the letters; cl100k: the upper-case-ish set)
    constexpr int TKB_QUAL = GEN ? -1 : (PAT == TK_PAT_O200K ? (int)TKB_L : (PAT == TK_PAT_CL100K ? (int)TKB_UP : -1));
    {
        uint32_t stopm = mk.start;
        if (!GEN) stopm &= ~tk_chunk_never(fam, st, prevc, near);
        stop16[tid] = (uint16_t)(stopm | cert);  // (a hard start -- a document begins -- is a start whatever the classes on its two sides)
        if constexpr (TKB_QUAL >= 0) ((uint16_t*)pool)[TKB_QUAL * (NW * 4) + tid] = (uint16_t)tk_chunk_second_stop(PAT, st, mk.text, cert, stopm & ~cert);
    }
    constexpr uint32_t T0 = TK2_LEFT / 16, T1 = (TK2_LEFT + TK_TILE) / 16;  // chunks [T0, T1) are the tile
    const bool in_tile = tid >= T0 && tid < T1;
    if (in_tile) ((uint16_t*)bits)[tid - T0] = (uint16_t)cert;
    // scan starts: the certain starts inside the tile (phase D takes them 64 positions at a time); plus the start of the piece that
    // crosses in from the left when the first char of the tile is not certain itself: the last certain start of the left context,
    // else a walk back through HBM
    uint32_t mine = in_tile ? cert : 0u;
    uint32_t extra = TKF_NONE;
    if (wid == 0) {
        // the last certain start of the left context (lanes 0 .. T0 - 1)
        const uint64_t lm = __ballot(cert != 0u) & ((1ull << T0) - 1ull);
        const int ll = lm ? 63 - __clzll((long long)lm) : 0;
        const uint32_t lc = (uint32_t)__shfl((int)cert, ll, 64);
        bool walk = false;
        if (tid == T0 && tile_start > 0 && tile_start < n) {
            const bool first_cert = mk.text && ((cert >> (__ffs((int)mk.text) - 1)) & 1u);
            if (!first_cert) {
                if (lm) extra = (uint32_t)ll * 16u + 31u - (uint32_t)__clz((int)lc);
                else walk = true;
            }
        }
#ifndef TKF_SYNC_POINTS
#define TKF_SYNC_POINTS 1
#endif
        if constexpr (TKF_SYNC_POINTS && (PAT == TK_PAT_O200K || PAT == TK_PAT_CL100K) && !SLOW) {
            // (round 6) No certain start in the left context -- the inside of a sentence of a script without spaces (a piece of 120 to 1000 bytes: half of the
            // tiles that went to the workgroup-wide scanner on web text).  A scan does not need a piece START to be in phase with the sequential regex, only a
            // position at which the matcher's state is known: a lower-case letter (LL) that follows a letter of class LL or LC, with no apostrophe in the three
            // bytes before it, is INSIDE a piece of the letter alternatives (no alternative ends between two such letters except behind a contraction), and
            // behind it the matcher is in the alternative's lower-case part -- exactly where a match that starts AT that letter is after its first char (the
            // optional prefix cannot take a letter, the upper-case part takes nothing): the piece ends where a scan from there ends.  A letter without case (LC,
            // in both parts) leaves one ambiguity -- an upper-case letter right behind the run of LC / mark chars that starts at it goes on in the upper-case
            // part and ends the lower-case one -- so it qualifies when that run ends inside the window and not at an upper-case letter.  The chain from such a
            // position records only true boundaries (its own start lies left of the tile).  Checked on 64 MiB of the bench corpus: 35.7 M such positions, the
            // scan from every one ends where its piece ends (tools/experiments/sync_points.cpp); 78 of the 84 tiles per 17 475 without a certain start have one.
            if (__ballot(walk) != 0ull) {  // (wave-uniform and rare: 0.5 % of the tiles)
                uint32_t s_ll = 0u, s_lc = 0u;
                if (tid < T0) {
                    if constexpr (PAT == TK_PAT_CL100K) {  // (cl100k: `\p{L}++` -- any letter behind a letter is inside it, and a scan from it ends where the run ends)
                        s_ll = tk_prev_set(st.l, TK_M_L, prevc) & mk.text & ~near & st.l;
                    } else {
                        const uint32_t after_letter = tk_prev_set(st.ll | st.lc, TK_CB(TK_C_LL) | TK_CB(TK_C_LC), prevc) & mk.text & ~near;
                        s_ll = after_letter & st.ll;
                        s_lc = after_letter & st.lc;
                    }
                }
                uint32_t at = TKF_NONE;
                const uint64_t b_ll = __ballot(s_ll != 0u), b_lc = __ballot(s_lc != 0u);
                if (b_ll) {
                    const int l2 = 63 - __clzll((long long)b_ll);
                    at = (uint32_t)l2 * 16u + 31u - (uint32_t)__clz((int)(uint32_t)__shfl((int)s_ll, l2, 64));
                } else if (b_lc) {
                    const int l2 = 63 - __clzll((long long)b_lc);
                    const uint32_t sp = (uint32_t)l2 * 16u + 31u - (uint32_t)__clz((int)(uint32_t)__shfl((int)s_lc, l2, 64));
                    // the end of the run of LC / mark bytes from there (the bitmaps are in LDS since the barrier above)
                    uint32_t j = (uint32_t)TK2_WIN;
                    for (uint32_t wq = sp >> 6; wq < (uint32_t)TK2_NSEG; ++wq) {
                        uint64_t v = ~bm[TKB_CAS][wq];
                        if (wq == (sp >> 6)) v &= ~0ull << (sp & 63u);
                        if (v) {
                            j = wq * 64u + (uint32_t)__ffsll((unsigned long long)v) - 1u;
                            break;
                        }
                    }
                    // (UP and not CAS: an upper-case letter.  A run that reaches the window's end is a piece that leaves the window: decided there, below)
                    // (... and an upper-case letter that begins a document ends the run like anything else: documents begin with one)
                    if (j >= (uint32_t)TK2_WIN || ((bm[TKB_HARD][j >> 6] >> (j & 63u)) & 1ull) || !((bm[TKB_UP][j >> 6] >> (j & 63u)) & 1ull)) at = sp;
                }
                if (walk && at != TKF_NONE) {
                    extra = at;
                    walk = false;
                }
            }
        }
        if (walk) need_walk = 1;
    }
    __syncthreads();
    if (!SLOW && need_walk) {  // no certain start in the left context: a tile for the workgroup-wide scanner
        // (and a note for the walk-backs of later tiles: is this tile one run of a single class without a certain start?)
        if (tid == T0) scan_sh[0] = tk_class_from_planes(mk.p, 0);
        __syncthreads();
        const uint32_t c0 = scan_sh[0];
        bool same = cert == 0u && valid == 0xFFFFu;
#pragma unroll
        for (int b = 0; b < 4; ++b) same = same && (mk.p[b] & 0xFFFFu) == (((c0 >> b) & 1u) ? 0xFFFFu : 0u);
        const bool wave_ok = __ballot(in_tile && !same) == 0ull;
        if (lane == 0) scan_sh[1 + wid] = wave_ok ? 1u : 0u;
        __syncthreads();
        if (tid == 0) out.tile_sum[tile] = (scan_sh[1] & scan_sh[2] & scan_sh[3] & scan_sh[4]) ? (uint8_t)c0 : (uint8_t)16;
        defer_tile();
        continue;
    }
    TKT(2);
    if (dbg & 0x4000) {  // (perf experiments: stop after this phase)
        if (tid == 0) out.tile_np[tile] = 0;
        continue;
    }
    // ---- D: one lane per scan start; only boundaries inside the tile are recorded
    const uint32_t* planes32 = (const uint32_t*)planes;
    TkWin2Acc acc{planes32, (const uint32_t*)bm[TKB_START], (const uint32_t*)bm[TKB_HARD], raw, base, n, false};
    const TkCoop coop{&T, text, n, brk, SPEC ? ss : nullptr, SPEC ? si : nullptr, btab, scan_sh, lastc, pat};  // (used when SLOW)
    // what the end e of the piece that starts at p means for this tile: TKF_CHAIN_END when the chain ends here (the piece reaches the tile
    // end -- its end is remembered for the probe -- or e is a certain start, which has its own scanner), else e: the scan goes on from
    // there (an uncertain boundary, recorded when it lies inside the tile)
    auto chain_step = [&](uint64_t p, uint64_t e) -> uint64_t {
        if (e >= tile_end) {  // (clamped to the 32-bit window offset)
            if (p < tile_end) {
                uint64_t rel = e - (uint64_t)base;
                last_end_sh = rel > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)rel;
            }
            return TKF_CHAIN_END;
        }
        if (e >= tile_start) {
            const uint32_t re = (uint32_t)((int64_t)e - base);  // inside the window (e < tile_end)
            if ((certw[re >> 5] >> (re & 31u)) & 1u) return TKF_CHAIN_END;
            atomicOr(&bits[(uint32_t)(e - tile_start) >> 5], 1u << ((uint32_t)(e - tile_start) & 31));
        }
        return e;
    };
    // One evaluation by one lane: the piece that starts at window position p - base.  A piece that leaves the window goes on the slow list
    // (answered by the whole workgroup below).
    auto piece_from = [&](uint64_t p) -> uint64_t {
        const uint32_t r = (uint32_t)((int64_t)p - base);
        uint32_t len = 0;
        if (r + 64u <= (uint32_t)TK2_WIN) {
            const uint32_t wi = r >> 6, sh = r & 63u;
            const uint32_t c = tk_class_at_lds(planes32, r);
            {  // most pieces are short: 32-position windows first (a third of the vector-ALU work of the 64-bit form)
                const TkWinLds32 w32((const uint32_t(*)[2 * NW])bm, r);
                len = tk_piece_len_bits32(w32, acc, p, c, pat);
            }
            if (len == 0) {
                const TkWinLds wl(bm, wi, sh);
                TkBmExt ext{bm, wi, sh, (uint32_t)TK2_WIN - r};
                len = tk_piece_len_bits(wl, acc, ext, p, c, pat);
            }
        }
        uint64_t e = p + len;
        if (len == 0) {
            acc.left = false;
            e = tk_piece_end_slow(&acc, p, pat);  // byte walk inside the window
            if (acc.left) {
                const uint32_t at = atomicAdd(&nslow_sh, 1u);
                if (at < TKF_SLOW_CAP) slowl[at] = (uint16_t)r;
                else if (SLOW) atomicOr(&out.counters[TK_CNT_ERR], 1u);  // (cannot happen: at most two pieces of a tile can leave its window)
                return TKF_CHAIN_END;
            }
        }
        if (e > n) e = n;
        return chain_step(p, e);
    };
    // The same for a piece of any length, anywhere, by the whole workgroup (uniform control flow): walks the chain from p until it ends
    // or re-enters the tile, where the lanes' scanners take over through the continuation list.
    bool gave_up = false;  // (SLOW: the walk towards the tile was longer than TKF_WALK_BUDGET windows)
    auto coop_chain = [&](uint64_t p) {
        if constexpr (SLOW) {
            const TkWalkLds walk{w2_raw, w2_planes, w2_start, w2_hard, w2_jump};
            uint32_t windows = 0;
            for (;;) {
                uint64_t e = p;
                if (fam != TK_PAT_R50K && pat.digits() && p < tile_start && (tk_class_byte_slow(&T, text, p, n, brk, coop.ss, coop.si) & 15u) == TK_C_NU)
                    e = tk_coop_skip_digit_groups(&coop, p, tile_start);  // whole three-digit groups left of the tile
                if (e == p && p + 1024u < tile_start) {  // far left of the tile: a window of pieces per step (nothing of them lies in this tile)
                    if ((dbg & TKF_DBG_MAY_GIVE_UP) && windows++ >= ((dbg & TKF_DBG_NO_BUDGET) ? 0u : TKF_WALK_BUDGET)) {  // (uniform: every lane counts the same)
                        gave_up = true;
                        return;
                    }
                    const uint64_t t = tk_coop_window_walk(&coop, &walk, p, tile_start);
                    if (t != p) {
                        p = t;
                        continue;
                    }
                }
                if (e == p) e = tk_coop_piece_end(&coop, p);
                const uint64_t nx = chain_step(p, e);  // (all threads compute the same; the bit and last_end updates are idempotent)
                if (nx == TKF_CHAIN_END) return;
                if (nx >= tile_start) {  // inside the tile (and the window) again: the lanes' scanners take over from the continuation list
                    __syncthreads();
                    const bool room = ncont_sh < CONT_CAP;
                    __syncthreads();
                    if (room) {
                        if (tid == 0) {
                            contl[ncont_sh] = (uint16_t)(nx - (uint64_t)base);
                            ncont_sh = ncont_sh + 1;
                        }
                        __syncthreads();
                        return;
                    }
                    // (the list is full -- a tile of nothing but uncertain boundaries: the workgroup walks on itself; chain_step records
                    // the boundaries and ends the chain at the next certain start or at the end of the tile)
                }
                p = nx;
            }
        }
    };
    // The tile starts inside a piece and its left context holds no certain start: find the last one before it and walk from there.
    if constexpr (SLOW) {
        if (need_walk) {
            uint64_t other;
            uint32_t crun;
            // Tiles before this one that are whole runs of the class of the byte before this tile, without a certain start (they were
            // deferred like this one and have said so in tile_sum), are jumped over, 256 tiles per step.
            const uint32_t cprev = (uint32_t)lastc[T0 - 1];
            uint64_t skipped = 0;
            for (;;) {
                const int64_t q = (int64_t)tile - 1 - (int64_t)skipped - (int64_t)tid;
                const bool match = q >= 0 && (uint32_t)out.tile_sum[q] == cprev;
                const uint32_t first = tk_coop_first(coop, match ? TKF_NONE : tid, false);
                if (first == TKF_NONE) {
                    skipped += 256;
                    continue;
                }
                skipped += first;
                break;
            }
            const uint64_t p0 = tk_coop_certain_before(&coop, tile_start - skipped * TK_TILE - 1, skipped ? cprev : 16u, &other, &crun);
            // A tile deep inside a long run of one class: everything from the first char after p0 up to the end of this window is that
            // class (and no hard start) -- the piece that covers the tile start began at p0 (or within a contraction's length of it) and
            // reaches beyond the window: there is no piece start in this tile.  Only the tile in which the run ends evaluates the piece.
            TkCoopAcc ca{coop};
            const bool stretch_uniform = other == TK_NO_POS || other < tk_next_char(ca, p0);
            const bool class_ok = crun >= (uint32_t)TK_C_NL && crun <= (uint32_t)TK_C_OT && (crun != TK_C_NU || fam == TK_PAT_R50K || pat.digits() == 0u);
            const uint32_t here = tid >= T0 ? (uint32_t)((tk_member16(mk.p, 1u << (crun & 15u)) == 0xFFFFu) & (mk.hard == 0u)) : 1u;
            const bool covered = stretch_uniform && class_ok && __syncthreads_and((int)here);
            if (!covered) coop_chain(p0);
        }
    }
    TKT(10);  // (deferred-tile instance: the walk back and the chain towards the tile)
    if (SLOW && gave_up) {  // on the second list: the tile runs again when the generic engine has split the chunk (tk_api.hip, stage_deferred)
        if (tid == 0) deferred[(n + TK_TILE - 1) / TK_TILE + 2 + atomicAdd(&out.counters[TK_CNT_DEFER2], 1u)] = (uint32_t)tile;
        // (round 6) ... and is left without piece starts: where the host has not waited for this kernel's counters (stage_deferred) the kernels behind
        // it run over the tile before the host learns that the batch has to be repeated -- what they read has to be a tile, if an empty one
        if (tid < TK_TILE / 32) {
            const uint64_t wgp = tile_start / 32 + tid;
            if (wgp * 32 < n) out.starts[wgp] = 0u;
        }
        if (tid == 0) out.tile_np[tile] = (uint32_t)(tile_end - tile_start) + (uint32_t)TK2_LEFT;
        continue;
    }
    // Round 0, for all the certain starts of the tile at once: between a piece start and the next position at which a piece MAY start
    // (`stop16`) there is no boundary, so a piece that starts at a certain start ends at the next stop whenever that stop is a certain
    // start too -- 85 % of the pieces, and nothing has to be recorded for them: the certain starts are in `bits` already.  A lane looks at
    // the 64 positions from its chunk on: adding the starts, shifted by one, to the complement of the stops carries each of them up
    // to its next stop.  Starts whose next stop is not certain (or lies further away) go on the continuation list.
    uint32_t own = 0u;           // starts of the lane that did not fit the list: evaluated by the lane itself
    bool own_extra = false;
    {
        const uint32_t te = (uint32_t)(tile_end - tile_start) + (uint32_t)TK2_LEFT;  // window position of the tile's end
        uint32_t sm = mine;                                                          // certain starts of the lane's chunk before the tile's end
        if (tid * 16u + 16u > te) sm &= tid * 16u >= te ? 0u : ((1u << (te - tid * 16u)) - 1u);
        uint32_t needm = 0u;  // starts that have to be evaluated
        if (sm) {
            const uint32_t* sw = (const uint32_t*)stop16;
            const uint32_t* cw = certw;
            const uint32_t wi = tid >> 1, sh = (tid & 1u) * 16u;
            const uint32_t s0 = sw[wi], s1 = sw[wi + 1], s2 = sw[wi + 2], c0 = cw[wi], c1 = cw[wi + 1], c2 = cw[wi + 2];  // (tid < T1: at most word 126 of 128)
            const uint64_t x = (uint64_t)__builtin_amdgcn_alignbit(s1, s0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(s2, s1, sh) << 32);
            const uint64_t cx = (uint64_t)__builtin_amdgcn_alignbit(c1, c0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(c2, c1, sh) << 32);
            const uint64_t gaps = ~x;
            const uint64_t sum = gaps + ((uint64_t)sm << 1);
            uint64_t nxt = sum & x;                 // the next stop of every start (a start is a stop: no carry passes one)
            if (sum < gaps) needm = 1u << (31 - __clz((int)sm));  // the carry of the last start left the 64 positions: no stop in sight
            uint64_t unc = nxt & ~cx;               // next stops that are not certain
            if constexpr (TKB_QUAL >= 0) {
                // ... of which some are known not to end the piece (a prefix char and its letter): the piece ends at the stop behind -- the same
                // addition once more, from those stops.  One whose next stop is certain is settled; the others stay "not certain", and the
                // start they are reached from (the last one below: there is no start between the two stops) is evaluated after all.
                const uint32_t* qw = (const uint32_t*)bm[TKB_QUAL];
                const uint32_t q0 = qw[wi], q1 = qw[wi + 1], q2 = qw[wi + 2];
                const uint64_t q = unc & ((uint64_t)__builtin_amdgcn_alignbit(q1, q0, sh) | ((uint64_t)(__builtin_amdgcn_alignbit(q2, q1, sh) & 0x7FFFFFFFu) << 32));  // (not the window's last position: nothing behind it is in sight)
                if (q) {
                    const uint64_t sum2 = gaps + (q << 1);
                    const uint64_t nxt2 = sum2 & x;
                    unc = (unc & ~q) | (nxt2 & ~cx);
                    if (sum2 < gaps) unc |= 1ull << (63 - __clzll((long long)q));  // (no stop in sight behind the last of them)
                    nxt |= nxt2;
                }
            }
            while (unc) {
                const uint32_t u = (uint32_t)__ffsll((unsigned long long)unc) - 1u;
                unc &= unc - 1ull;
                const uint32_t below = sm & (u >= 16u ? 0xFFFFu : ((1u << u) - 1u));
                needm |= 1u << (31 - __clz((int)below));  // (the stop was reached from the last start below it)
            }
            // the tile's last piece ends at a certain start at or behind the tile's end: its end is remembered for the probe
            const uint32_t k = te - tid * 16u;  // (sm != 0: the chunk starts before the tile's end)
            const uint64_t ce = nxt & cx & (k < 64u ? ~((1ull << k) - 1ull) : 0ull);
            if (ce) last_end_sh = tid * 16u + (uint32_t)__ffsll((unsigned long long)ce) - 1u;
        }
        const uint32_t cnt = (uint32_t)__popc(needm) + ((tid == T0 && extra != TKF_NONE) ? 1u : 0u);
        if (cnt) {
            uint32_t at = atomicAdd(&ncont_sh, cnt);
            if (tid == T0 && extra != TKF_NONE) {
                if (at < CONT_CAP) contl[at++] = (uint16_t)extra;
                else own_extra = true;
            }
            for (uint32_t m = needm; m; m &= m - 1) {
                if (at < CONT_CAP) contl[at++] = (uint16_t)(tid * 16u + (uint32_t)__ffs((int)m) - 1u);
                else own |= m & (0u - m);
            }
        }
    }
    TKT(12);  // (round 0 without the barrier)
    __syncthreads();
    TKT(11);  // (round 0 and its barrier)
    // Rounds around ONE instance of the lanes' evaluation (the scanner is big: a second inlined copy spills registers): the listed
    // chains (and what did not fit the list: every lane's own), walked to their ends.  After each round the workgroup answers the pieces
    // that left the window; those can add continuations for one more round.
    uint32_t cont_done = 0, slow_done = 0;
    for (int round = 1;; ++round) {
        const uint32_t cont_n = ncont_sh < CONT_CAP ? ncont_sh : CONT_CAP;
        const uint32_t lo = cont_done, hi = cont_n;
        cont_done = cont_n;
        uint32_t i = lo + tid;
        for (;;) {
            uint64_t p = TKF_CHAIN_END;
            if (i < hi) {
                p = (uint64_t)(base + contl[i]);
                i += 256;
            } else if (round == 1) {
                if (own_extra) {
                    own_extra = false;
                    p = (uint64_t)(base + extra);
                } else if (own) {
                    p = (uint64_t)(base + (int64_t)(tid * 16u + (uint32_t)__ffs((int)own) - 1u));
                    own &= own - 1;
                }
            }
            if (!__any(p != TKF_CHAIN_END)) break;
            for (;;) {
                const uint64_t e = p != TKF_CHAIN_END ? piece_from(p) : TKF_CHAIN_END;
                p = e;
                if (!__any(p != TKF_CHAIN_END)) break;
            }
        }
        TKT(13);  // (the evaluations of a round)
        __syncthreads();
        if (!SLOW && nslow_sh) break;  // a piece leaves the window: the tile is deferred (below)
        const uint32_t slow_n = nslow_sh < TKF_SLOW_CAP ? nslow_sh : (uint32_t)TKF_SLOW_CAP;
        for (uint32_t q = slow_done; q < slow_n; ++q) coop_chain((uint64_t)(base + slowl[q]));
        slow_done = slow_n;
        __syncthreads();
        const uint32_t cont_now = ncont_sh < CONT_CAP ? ncont_sh : CONT_CAP;
        if (cont_now == cont_done) break;
    }
    if (!SLOW && nslow_sh) {
#ifndef TKF_EXTEND
#define TKF_EXTEND 1
#endif
        // (round 6) A piece leaves the window.  The other half of the tiles that went to the workgroup-wide scanner on web text: a sentence of a script without
        // spaces (120 to 1000 bytes, one piece) that starts near the tile's end.  Everything else about the tile is known -- what is missing is where that piece
        // ends, and for a letter piece that runs into the window's end inside a run of letters of ONE kind that is the end of the run: cl100k (`\p{L}++`) any
        // letters, ended by whatever is not a letter; o200k letters without case and marks (LC, MK: in the upper-case AND the lower-case part of its
        // alternatives, so the matcher's state does not matter) and lower-case letters (LL: from the first of them on the matcher is in the lower-case part),
        // ended by anything but an apostrophe (a contraction may follow) or an upper-case letter before the first lower-case one (the piece may go on): then
        // the tile is deferred as before.  One wavefront reads on from the window's end, eight bytes per lane and 512 per step, a class lookup per char, up to
        // 2 KiB.  The piece must be a letter piece: it starts with a letter, or with one char of a prefix class whose next char is a letter.  Checked on the
        // bench corpora (tools/experiments/extend_piece.cpp): applies to 78 of 78 such tiles per 17 474 (o200k web text) and 23 of 23 (cl100k mixed text), right
        // every time.
        bool extended = false;
        if constexpr (TKF_EXTEND && (PAT == TK_PAT_O200K || PAT == TK_PAT_CL100K)) {
            constexpr bool CL = PAT == TK_PAT_CL100K;
            constexpr uint32_t SET = CL ? TK_M_L : (TK_CB(TK_C_LC) | TK_CB(TK_C_MK) | TK_CB(TK_C_LL));
            constexpr uint32_t LETTER = CL ? TK_M_L : (TK_M_L | TK_CB(TK_C_MK));
            constexpr uint32_t PREFIX = TK_CB(TK_C_SP) | TK_CB(TK_C_WSO) | TK_CB(TK_C_SL) | TK_CB(TK_C_OT) | (CL ? TK_CB(TK_C_MK) : 0u);
            if (wid == 0) {
                uint32_t end_rel = 0u;  // the piece's end, relative to the window's base (0: the rule does not apply)
                const uint32_t* st32 = (const uint32_t*)bm[TKB_START];
                const uint32_t r = slowl[0];
                const uint32_t lastw = st32[TK2_WIN / 32 - 1];
                bool ok = nslow_sh == 1u && lastw != 0u && (uint64_t)(base + TK2_WIN) < n;
                bool last_ll = false;
                if (ok) {
                    const uint32_t cp = tk_class_at_lds(planes32, r);
                    uint32_t x = (st32[r >> 5] >> (r & 31u)) >> 1;
                    const uint32_t nx = x ? r + (uint32_t)__ffs((int)x) : ((r >> 5) + 1u) * 32u + (uint32_t)__ffs((int)st32[(r >> 5) + 1u]) - 1u;  // the next char start
                    const uint32_t cn = nx < (uint32_t)TK2_WIN ? tk_class_at_lds(planes32, nx) : 0u;
                    const bool hard_nx = nx < (uint32_t)TK2_WIN && ((((const uint32_t*)bm[TKB_HARD])[nx >> 5] >> (nx & 31u)) & 1u);
                    const uint32_t lastc = tk_class_at_lds(planes32, (uint32_t)TK2_WIN - 32u + 31u - (uint32_t)__clz((int)lastw));
                    ok = (((LETTER >> cp) & 1u) || (((PREFIX >> cp) & 1u) && ((LETTER >> cn) & 1u) && !hard_nx)) && ((SET >> lastc) & 1u);
                    last_ll = !CL && lastc == (uint32_t)TK_C_LL;
                }
                if (ok) end_rel = tk_extend_letter_run<CL>(TKF_PTR(const uint8_t*, ext_sh[1]), TKF_PTR(const uint8_t*, ext_sh[2]), TKF_PTR(const uint8_t*, ext_sh[3]), text, n, TKF_PTR(const uint32_t*, ext_sh[0]), (uint64_t)(base + TK2_WIN), lane, last_ll);
                if (lane == 0) {
                    scan_sh[0] = end_rel;
                    if (end_rel) last_end_sh = end_rel;
                }
            }
            __syncthreads();
            extended = scan_sh[0] != 0u;
        }
        if (!extended) {
            defer_tile();
            continue;
        }
    }
    }
    TKT(3);
    if (dbg & 0x8000) {  // (perf experiments: stop after this phase)
        if (tid == 0) out.tile_np[tile] = 0;
        continue;
    }
    if constexpr (SLOW) {  // the starts are found: the rest is the other instance's (TKF_MODE_GIVEN)
        if (tid < TK_TILE / 32) {
            const uint64_t wgp = tile_start / 32 + tid;
            if (wgp * 32 < n) out.starts[wgp] = bits[tid];
        }
        if (tid == 0) out.tile_np[tile] = last_end_sh;
        continue;
    } else {
    // ---- E (round 6): the tile's pieces by length class, straight from the start bitmap.  Rounds 2-5 wrote a list of piece starts (a
    // lane per bitmap word, a store per set bit) and then classified every piece from it -- two reads of the list, four ballots and
    // their prefix counts per row of 256 pieces: 1240 of the kernel's 8200 vector instructions per tile.  Here ONE wavefront (the other three
    // wait at the barrier: vector instructions are paid per wavefront) takes two words per lane and derives the classes as bit
    // operations: "a start follows within d positions" = the bitmap OR-ed with its own shifts (doubling: 1-2, 1-4, 1-8, 1-16, 16-23),
    // so a piece of at most 4 / 8 / TK_XL_MAX bytes is a set bit under the mask of that distance and under no shorter one; counts by
    // v_bcnt, places by two wave scans, and a lane then writes the positions of ITS set bits of one class after the other.  A list
    // entry is the piece's window position (its length is the distance to the next set bit: one funnel shift and a count of trailing
    // zeros where the row needs it), beside it the piece's index k in the tile (its result word is res[run + k]).
    constexpr uint32_t NWB = (uint32_t)TK_TILE / 32u;  // words of the tile's bitmap (even)
    const uint32_t run_base = (uint32_t)tile * TKF_CAP;
#if TKF_PARK_ARGS
    // (the names of the kernel's arguments, shadowed by what was parked: the code below reads as before)
#if TKF_PARK_ARGS == 1
    auto unpark = [&](int i) -> uint64_t {
        const uint64_t v = park_sh[i];
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
    };
#endif
    TkTables T_f{};
    TkFrontOut out_f{};
    TkMissKey* __restrict__ mt_f = nullptr;
    uint32_t mt_mask_f = 0;
#if TKF_PARK_ARGS == 2
    // The kernarg segment once more: a pointer the compiler cannot see through, in the constant address space (scalar loads).  What the rows and the dense
    // passes need of the tables is loaded WHERE IT IS USED (fresh_T(): the pointer goes through an empty asm statement every time, so the loads are not
    // moved out of the loop of the rows -- which uses a different table in each of its branches: all of them alive across the whole loop were spilled
    // scalar registers reloaded by vector instructions in every branch); what is used everywhere (the result words' base) once, here.
    typedef const __attribute__((address_space(4))) uint8_t* KArg;
    KArg ka_e = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka_e));
#define TKF_KA64(ka, off) (*(const __attribute__((address_space(4))) uint64_t*)((ka) + (off)))
#define TKF_KA32(ka, off) (*(const __attribute__((address_space(4))) uint32_t*)((ka) + (off)))
#define TKF_KA_TP(ka, member) TKF_PTR(decltype(TkTables::member), TKF_KA64(ka, offsetof(TkFrontArgs, T) + offsetof(TkTables, member)))
#define TKF_KA_T32(ka, member) TKF_KA32(ka, offsetof(TkFrontArgs, T) + offsetof(TkTables, member))
#define TKF_KA_OP(ka, type, member) TKF_PTR(type, TKF_KA64(ka, offsetof(TkFrontArgs, out) + offsetof(TkFrontOut, member)))
    auto fresh_T = [&]() -> TkTables {
        KArg ka = ka_e;
        asm volatile("" : "+s"(ka));
        TkTables t{};
        t.short_tab = TKF_KA_TP(ka, short_tab); t.short_mask = TKF_KA_T32(ka, short_mask); t.short_shift = TKF_KA_T32(ka, short_shift);
        t.mid_tab = TKF_KA_TP(ka, mid_tab); t.mid_mask = TKF_KA_T32(ka, mid_mask); t.mid_shift = TKF_KA_T32(ka, mid_shift);
        t.xl = TKF_KA_TP(ka, xl); t.xl_mask = TKF_KA_T32(ka, xl_mask); t.max_token_len = TKF_KA_T32(ka, max_token_len);
        t.tok_bytes = TKF_KA_TP(ka, tok_bytes); t.piece = TKF_KA_TP(ka, piece); t.piece_off = TKF_KA_TP(ka, piece_off);
        t.piece_mask = TKF_KA64(ka, offsetof(TkFrontArgs, T) + offsetof(TkTables, piece_mask));
        if constexpr (SPEC) {
            t.n_spec = TKF_KA_T32(ka, n_spec); t.spec_bytes = TKF_KA_TP(ka, spec_bytes); t.spec_id = TKF_KA_TP(ka, spec_id); t.spec_off = TKF_KA_TP(ka, spec_off);
        }
        return t;  // (the loads of the fields a caller does not use are dead code)
    };
    auto fresh_out = [&]() -> TkFrontOut {  // (the rare ways: a claim that succeeds, an overflow entry)
        KArg ka = ka_e;
        asm volatile("" : "+s"(ka));
        TkFrontOut o{};
        o.data.tab = TKF_PTR(TkMissTab*, TKF_KA64(ka, offsetof(TkFrontArgs, out) + offsetof(TkFrontOut, data) + offsetof(TkMiss, tab)));
        o.data.ovf = TKF_PTR(TkMissOvf*, TKF_KA64(ka, offsetof(TkFrontArgs, out) + offsetof(TkFrontOut, data) + offsetof(TkMiss, ovf)));
        o.data.ovf_base = TKF_KA32(ka, offsetof(TkFrontArgs, out) + offsetof(TkFrontOut, data) + offsetof(TkMiss, ovf_base));
        o.ovf_cap = TKF_KA32(ka, offsetof(TkFrontArgs, out) + offsetof(TkFrontOut, ovf_cap));
        o.listC = TKF_KA_OP(ka, uint32_t*, listC);
        o.counters = TKF_KA_OP(ka, uint32_t*, counters);
        return o;
    };
    auto fresh_mt = [&](uint32_t& mask_out) -> TkMissKey* {
        KArg ka = ka_e;
        asm volatile("" : "+s"(ka));
        mask_out = TKF_KA32(ka, offsetof(TkFrontArgs, mt_mask));
        return TKF_PTR(TkMissKey*, TKF_KA64(ka, offsetof(TkFrontArgs, mt)));
    };
    if constexpr (!SLOW) {
        out_f.starts = TKF_KA_OP(ka_e, uint32_t*, starts);
        out_f.tile_np = TKF_KA_OP(ka_e, uint32_t*, tile_np);
        out_f.res = TKF_KA_OP(ka_e, uint32_t*, res);
        mt_f = fresh_mt(mt_mask_f);
    }
#else
    if constexpr (!SLOW) {
        T_f.short_tab = (const TkShortSlot*)(uintptr_t)unpark(0);
        T_f.mid_tab = (const TkPieceSlot*)(uintptr_t)unpark(1);
        T_f.xl = (const TkXlSlot*)(uintptr_t)unpark(2);
        T_f.tok_bytes = (const uint8_t*)(uintptr_t)unpark(3);
        mt_f = (TkMissKey*)(uintptr_t)unpark(4);
        out_f.res = (uint32_t*)(uintptr_t)unpark(5);
        out_f.data.tab = (TkMissTab*)(uintptr_t)unpark(6);
        out_f.data.ovf = (TkMissOvf*)(uintptr_t)unpark(7);
        out_f.listC = (uint32_t*)(uintptr_t)unpark(8);
        const uint64_t a = unpark(9), b = unpark(10), c = unpark(11), d = unpark(12);
        T_f.short_mask = (uint32_t)a; T_f.short_shift = (uint32_t)(a >> 32);
        T_f.mid_mask = (uint32_t)b; T_f.mid_shift = (uint32_t)(b >> 32);
        T_f.xl_mask = (uint32_t)c; T_f.max_token_len = (uint32_t)(c >> 32);
        mt_mask_f = (uint32_t)d; out_f.data.ovf_base = (uint32_t)(d >> 32);
        out_f.ovf_cap = (uint32_t)unpark(13);
        T_f.piece = (const TkPieceSlot*)(uintptr_t)unpark(14);
        T_f.piece_off = (const uint32_t*)(uintptr_t)unpark(15);
        T_f.piece_mask = unpark(16);
        out_f.starts = out.starts; out_f.tile_np = out.tile_np; out_f.counters = out.counters;
        if constexpr (SPEC) {  // (the instances with allowed special tokens look their ids up here: tk_special_id)
            T_f.n_spec = T.n_spec; T_f.spec_bytes = T.spec_bytes; T_f.spec_id = T.spec_id; T_f.spec_off = T.spec_off;
        }
    }
#endif
    const TkTables& T_outer = T;
    (void)T_outer;
#if TKF_PARK_ARGS == 2
    (void)T_f;
#define T fresh_T()  /* (undefined again where the kernel's loop ends) */
#if TKF_FRESH_SCALARS
    // The same for the tests of the debug word's bits in phases E and F: each was a 64-bit lane mask computed before the loop of the rows, spilled, and
    // reloaded by two vector instructions in every row and every dense pass.  Through an empty asm statement they are a scalar test at the point of use.
    auto dbg_fresh = [&]() -> int { int d = dbg; asm volatile("" : "+s"(d)); return d; };
#define dbg dbg_fresh()
#if TKF_FRESH_SCALARS == 1
    auto tile_start_fresh = [&]() -> uint64_t { uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tile); asm volatile("" : "+s"(t)); return (uint64_t)t * (uint64_t)TK_TILE; };
    auto base_fresh = [&]() -> int64_t { return (int64_t)tile_start_fresh() - TK2_LEFT; };
#define tile_start tile_start_fresh()
#define base base_fresh()
#endif
#endif
#else
    const TkTables& T = T_f;
#endif
    const TkFrontOut& out = out_f;
    TkMissKey* __restrict__ mt = mt_f;
    uint32_t mt_mask = mt_mask_f;
#endif
    uint32_t* bx = certw;                            // [128] the start bitmap once more, plus the END of the tile's last piece (the certain starts are dead)
    uint16_t* ord_sl = (uint16_t*)btab;              // [1024] short pieces from the front, long ones from the back: window positions (the byte table is dead)
    uint16_t* ord_m = (uint16_t*)planes;             // [1024] mid pieces (the planes are dead)
    uint16_t* ordk_sl = (uint16_t*)pool;             // [1024] their indices k (the bitmaps are dead)
    uint16_t* ordk_m = (uint16_t*)(pool + 2048);     // [1024]
    uint32_t* mlist = (uint32_t*)(pool + 4096);      // [TKF_BATCH] pieces that are not tokens: position | k << 12 | class << 24 (3: more than TK_XL_MAX bytes) | "no slot" << 31
    uint32_t& ntail_sh = ncont_sh;   // pieces of the tile that are not tokens, so far = entries at the tail of its run (the scanners' counter is dead)
    if (wid == 0) {
        const uint32_t i0 = 2u * (uint32_t)lane, last_end0 = last_end_sh;
        const uint32_t v0 = i0 < NWB ? bits[i0] : 0u, v1 = i0 < NWB ? bits[i0 + 1u] : 0u;
        const uint32_t le = last_end0 - (uint32_t)TK2_LEFT;  // (tile-relative; beyond the bitmap when the last piece leaves the window: a tile of the deferred list)
        bx[i0] = v0 | ((le >> 5) == i0 ? 1u << (le & 31u) : 0u);
        bx[i0 + 1u] = v1 | ((le >> 5) == i0 + 1u ? 1u << (le & 31u) : 0u);
        const uint32_t inc = tk_wave_scan_u32((uint32_t)__popc(v0) + (uint32_t)__popc(v1), lane);
        const uint64_t wgp = tile_start / 32 + i0;
        if (i0 < NWB && wgp * 32 < n) *(uint2*)(out.starts + wgp) = make_uint2(v0, v1);  // (8-byte aligned: 120 words per tile; the array has two words to spare)
        if (lane == 63) {
            np_sh = inc;
            out.tile_np[tile] = inc;
        }
        if (lane == 0) {
            ntail_sh = 0;
            if (GEN) ngap_sh = 0;
        }
    }
    if constexpr (SPEC) {  // starts of special tokens in the window, for the lists (the break bitmap is dead since phase C)
        if (tid < TK2_WIN / 32) {
            const int64_t wgp = base + (int64_t)tid * 32;
            brkw[tid] = (wgp >= 0 && (uint64_t)wgp < n) ? ss[wgp >> 5] : 0u;
        }
    }
    __syncthreads();
    TKT(4);
    if (dbg & 0x10000) {  // (perf experiments: stop after this phase)
        if (tid == 0) out.tile_np[tile] = 0;
        continue;
    }
    const uint32_t np = np_sh, last_end = last_end_sh;
    const uint32_t* dwr = (const uint32_t*)raw;
    const bool short_tab = T.short_tab != nullptr;
    // length of the piece at window position pos when the next start lies within 31 positions (every piece of the class lists)
    auto near_len = [&](uint32_t pos) -> uint32_t {
        const uint32_t t = pos - (uint32_t)TK2_LEFT, wi = t >> 5;
        return (uint32_t)__ffs((int)(__builtin_amdgcn_alignbit(bx[wi + 1u], bx[wi], t & 31u) >> 1));
    };
    // end (window position) of the piece that starts at pos, however long
    auto far_end = [&](uint32_t pos) -> uint32_t {
        const uint32_t t = pos - (uint32_t)TK2_LEFT;
        uint32_t wi = t >> 5;
        uint32_t x = (bx[wi] >> (t & 31u)) >> 1;
        if (x) return pos + (uint32_t)__ffs((int)x);
        for (++wi; wi < (uint32_t)TK2_WIN / 32u; ++wi) {
            x = bx[wi];
            if (x) return (uint32_t)TK2_LEFT + wi * 32u + (uint32_t)__ffs((int)x) - 1u;
        }
        return last_end;
    };
    // A piece that is not a token, by its identity (w0, w1, w2; kk = tk_ident_hash): claim a slot of the in-call table (first occurrence:
    // the slot's entry gets the piece, the merge kernels will find it there) or find it claimed by IDENTICAL bytes; `k0` / `k1` = the two
    // halves of slot i, which the caller has loaded.  Returns the slot, or TKF_NONE when the neighbourhood is full.  Slots are written
    // once, so a cached load can only be stale towards "empty" / "not written yet", where the atomic (the load at the memory side) decides.
    auto claim = [&](uint64_t w0, uint64_t w1, uint64_t w2, unsigned long long kk, bool exact, bool in_lds, uint32_t s_loc, uint64_t gs, uint32_t len,
                     uint32_t i, ulonglong2 k0, ulonglong2 k1) -> uint32_t {
        for (int p = 0;;) {
            unsigned long long cur = k0.x;
            if (cur == TK_EMPTY_KEY) cur = atomicCAS(&mt[i].key, TK_EMPTY_KEY, kk);
            if (cur == TK_EMPTY_KEY) {  // claimed: this occurrence is the one that gets merged
                __hip_atomic_store(&mt[i].w0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&mt[i].w1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&mt[i].w2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if TKF_PARK_ARGS == 2
                *(uint2*)&fresh_out().data.tab[i].start = make_uint2((uint32_t)gs, len);
#else
                *(uint2*)&out.data.tab[i].start = make_uint2((uint32_t)gs, len);
#endif
                return i;
            }
            if (cur == kk) {
                unsigned long long a0 = k0.y, a1 = k1.x, a2 = k1.y;
                if (a2 == TK_EMPTY_KEY || a0 == TK_EMPTY_KEY || a1 == TK_EMPTY_KEY) {  // (a line cached before the claimant had written: once more, at the memory side)
                    // The claimant writes its words right behind its compare-and-swap: whoever loses that race by a few hundred nanoseconds --
                    // every tile of a text that repeats itself reaches its first missed piece at the same moment -- would find them empty, take
                    // the piece for another one and claim the next slot for the same bytes (exact, but a merge per slot, and a full
                    // neighbourhood sends the piece to the overflow entries).  So it looks again a few times (bounded: never a dead lock).
                    for (int spin = 0;; ++spin) {
                        a0 = __hip_atomic_load(&mt[i].w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a1 = __hip_atomic_load(&mt[i].w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a2 = __hip_atomic_load(&mt[i].w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((a2 != TK_EMPTY_KEY && a0 != TK_EMPTY_KEY && a1 != TK_EMPTY_KEY) || spin >= TKF_CLAIM_SPIN) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                bool same = a0 == w0 && a1 == w1 && a2 != TK_EMPTY_KEY && (exact ? a2 == w2 : (a2 >> 32) == (w2 >> 32));
                if (same && !exact)
                    same = in_lds ? tk_equal_lds_text(raw, s_loc + 8u, text, (uint64_t)(uint32_t)a2 + 8u, len - 16u)  // (w0 and w1 are the first and the last eight bytes)
                                  : tk_equal_bytes(text, gs, text, (uint32_t)a2, len);
                if (same) return i;
            }
            if (++p == TK_MT_PROBES) return TKF_NONE;
            i = (i + 1) & mt_mask;
            k0 = *(const ulonglong2*)&mt[i].key;
            k1 = *(const ulonglong2*)&mt[i].w1;
        }
    };
    // the result of a piece that is not a token: its word, and its entry once more at the tail of the run (one LDS atomic per wavefront)
    auto put_ref = [&](bool on, uint32_t k, uint32_t ref) {
        const uint64_t m = __ballot(on);
        if (m) {
            uint32_t at = 0;
            const int leader = __ffsll((unsigned long long)m) - 1;
            if (lane == leader) at = atomicAdd(&ntail_sh, (uint32_t)__popcll(m));
            at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (on) {
                out.res[run_base + k] = ref != TKF_NONE ? (TK_RES_FLAG | ref) : 0u;
                out.res[run_base + TKF_TAIL_REFS - at] = ref;  // (TKF_NONE = 0xFFFFFFFF: counts as the one token 0 the result word holds; the batch is repeated with more room)
            }
        }
    };
    // A tile of more than TKF_BATCH pieces (lists of 1024 entries; rare: a table of digits, a page of one-letter words) goes through the lists
    // in parts of TKF_BL lanes = 28 bitmap words = 896 positions each.
    const uint32_t nbatch = np <= (uint32_t)TKF_BATCH ? 1u : (NWB / 2u + (uint32_t)TKF_BL - 1u) / (uint32_t)TKF_BL;
    // The pieces that are not tokens: a list per WAVEFRONT (no barrier between the rows and their claims, no atomic to append), drained in
    // dense passes of sixty-four -- when the wavefront's rows are done, or before a row that might not fit.
    constexpr uint32_t MLW = (uint32_t)TKF_BATCH / 4u;  // entries of a wavefront's list
    uint32_t* mlw = mlist + (uint32_t)wid * MLW;
    for (uint32_t bt = 0; bt < nbatch; ++bt) {
        // ---- the lists.  Every wavefront derives the masks (two words per lane) and the places; wavefront 0 then writes the short pieces,
        // 1 the mid ones, 2 the long ones, 3 those of more than TK_XL_MAX bytes (and special tokens, gap chars): the serial part -- a lane
        // writes its set bits one after the other -- is as long as the longest of the four, not as their sum.
        uint32_t n_mine = 0;  // entries of this wavefront's list of the pieces that are not tokens
        {
            // (the lane number as a value the compiler cannot see through: it would compute the lane's LDS addresses once, before this loop that
            // runs once, keep them across the whole of it -- and, out of registers, put them in scratch memory: four dependent reloads here)
            uint32_t lane_e = (uint32_t)lane;
            asm volatile("" : "+v"(lane_e));
            const uint32_t i0 = 2u * lane_e;
            const bool inb = nbatch == 1u || (lane_e >= bt * (uint32_t)TKF_BL && lane_e < (bt + 1u) * (uint32_t)TKF_BL);
            const uint32_t q0 = i0 < NWB ? bits[i0] : 0u, q1 = i0 < NWB ? bits[i0 + 1u] : 0u;  // the lane's pieces (all of them count towards k)
            const uint32_t cq = (uint32_t)__popc(q0) + (uint32_t)__popc(q1);
            const uint32_t kb0 = tk_wave_scan_u32(cq, lane) - cq, kb1 = kb0 + (uint32_t)__popc(q0);  // k of the first piece of either word
            const uint32_t x1 = bx[i0 + 1u], x2 = bx[i0 + 2u < (uint32_t)TK2_WIN / 32u ? i0 + 2u : i0];
            uint32_t p0 = inb ? q0 : 0u, p1 = inb ? q1 : 0u;
            const uint32_t posb = (uint32_t)TK2_LEFT + 32u * i0;  // window position of bit 0 of the lane's first word
            auto each = [&](uint32_t m, uint32_t qw, uint32_t kb, uint32_t pb, auto&& put) {
                while (m) {
                    const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                    m &= m - 1u;
                    put(pb + b, kb + (uint32_t)__popc(qw & ((1u << b) - 1u)));
                }
            };
            if constexpr (SPEC) {  // a special token: its id (src/lib.rs:426-434)
                const uint32_t s0 = i0 < NWB ? p0 & brkw[4u + i0] : 0u, s1 = i0 < NWB ? p1 & brkw[5u + i0] : 0u;
                if (wid == 3 && __ballot((s0 | s1) != 0u)) {
                    auto put = [&](uint32_t pos, uint32_t k) { out.res[run_base + k] = tk_special_id(T, text, (uint64_t)(base + pos), far_end(pos) - pos); };
                    each(s0, q0, kb0, posb, put);
                    each(s1, q1, kb1, posb + 32u, put);
                }
                p0 &= ~s0;
                p1 &= ~s1;
            }
            if constexpr (GEN) {  // a gap char (a char at which a pat_str of the generic engine matches nothing): no token
                if (gapb) {
                    const uint64_t wgp = tile_start / 32 + i0;
                    const uint32_t g0 = (i0 < NWB && wgp * 32 < n) ? p0 & gapb[wgp] : 0u, g1 = (i0 < NWB && (wgp + 1) * 32 < n) ? p1 & gapb[wgp + 1] : 0u;
                    if (wid == 3 && __ballot((g0 | g1) != 0u)) {
                        auto put = [&](uint32_t, uint32_t k) { out.res[run_base + k] = TK_RES_GAP; };
                        each(g0, q0, kb0, posb, put);
                        each(g1, q1, kb1, posb + 32u, put);
                        if (g0 | g1) atomicAdd(&ngap_sh, (uint32_t)__popc(g0) + (uint32_t)__popc(g1));  // (LDS; gap chars are rare)
                    }
                    p0 &= ~g0;
                    p1 &= ~g1;
                }
            }
            // pieces of the word A (B = the 32 positions behind it) whose next start is at most 4 / 8 / TK_XL_MAX positions away
            uint32_t n4a, n8a, n23a, n4b, n8b, n23b;
            auto reach = [&](uint32_t A, uint32_t B, uint32_t& n4, uint32_t& n8, uint32_t& n23) {
                const uint32_t s1 = __builtin_amdgcn_alignbit(B, A, 1), b1 = B >> 1;                       // distance 1
                const uint32_t n2l = s1 | __builtin_amdgcn_alignbit(b1, s1, 1), n2h = b1 | (b1 >> 1);     // 1 .. 2
                const uint32_t n4l = n2l | __builtin_amdgcn_alignbit(n2h, n2l, 2), n4h = n2h | (n2h >> 2);  // 1 .. 4
                const uint32_t n8l = n4l | __builtin_amdgcn_alignbit(n4h, n4l, 4), n8h = n4h | (n4h >> 4);  // 1 .. 8
                const uint32_t n16l = n8l | __builtin_amdgcn_alignbit(n8h, n8l, 8);                        // 1 .. 16
                static_assert(TK_XL_MAX == 23u && TK_XL_MIN == 9u, "the doubling ends at 23");
                n4 = n4l;
                n8 = n8l;
                n23 = n16l | __builtin_amdgcn_alignbit(n8h, n8l, 15);                                      // 1 .. 23
            };
            reach(bx[i0], x1, n4a, n8a, n23a);
            reach(x1, x2, n4b, n8b, n23b);
            // this wavefront's class: its pieces of the two words, their number, their places
            const uint32_t c0 = wid == 0 ? p0 & n4a : (wid == 1 ? p0 & n8a & ~n4a : (wid == 2 ? p0 & n23a & ~n8a : p0 & ~n23a));
            const uint32_t c1 = wid == 0 ? p1 & n4b : (wid == 1 ? p1 & n8b & ~n4b : (wid == 2 ? p1 & n23b & ~n8b : p1 & ~n23b));
            const uint32_t nc = (uint32_t)__popc(c0) + (uint32_t)__popc(c1);
            const uint32_t inc = tk_wave_scan_u32(nc, lane);
            if (lane == 63) scan_sh[wid] = inc;  // (n_s, n_m, n_l, n_xl)
            uint32_t o = inc - nc;
            if (wid == 0) {
                auto put = [&](uint32_t pos, uint32_t k) { ord_sl[o] = (uint16_t)pos; ordk_sl[o] = (uint16_t)k; ++o; };
                each(c0, q0, kb0, posb, put);
                each(c1, q1, kb1, posb + 32u, put);
            } else if (wid == 1) {
                auto put = [&](uint32_t pos, uint32_t k) { ord_m[o] = (uint16_t)pos; ordk_m[o] = (uint16_t)k; ++o; };
                each(c0, q0, kb0, posb, put);
                each(c1, q1, kb1, posb + 32u, put);
            } else if (wid == 2) {
                auto put = [&](uint32_t pos, uint32_t k) { ord_sl[1023u - o] = (uint16_t)pos; ordk_sl[1023u - o] = (uint16_t)k; ++o; };
                each(c0, q0, kb0, posb, put);
                each(c1, q1, kb1, posb + 32u, put);
            } else {
                // the pieces of more than TK_XL_MAX bytes are on the lists of the pieces that are not tokens from the start (the j-th on the list
                // of wavefront j & 3): no look at the vocabulary (2 % of the pieces, next to none of them tokens; tk_k_bincount asks once per DISTINCT piece)
                auto put = [&](uint32_t pos, uint32_t k) { mlist[(o & 3u) * MLW + (o >> 2)] = pos | (k << 12) | (3u << 24); ++o; };
                each(c0, q0, kb0, posb, put);
                each(c1, q1, kb1, posb + 32u, put);
            }
        }
        TKT(5);
        __syncthreads();
        TKT(6);
        // ---- F: whole-piece probe (src/lib.rs:367) by rows of 64 pieces of ONE length class: a wavefront takes every fourth row -- the
        // dearest rows first, so that they are spread evenly and run side by side -- and runs that class's probe only, with all lanes busy:
        //   long : identity = the bytes (tk_common.h, tk_ident), 32-byte slots that hold them
        //   mid  : 64-bit key = the bytes, 16-byte slots
        //   short: the bytes are the key (one unaligned LDS dword), 8-byte slots
        // A piece that is not a token goes on the wavefront's list.
        const uint32_t n_s = scan_sh[0], n_m = scan_sh[1], n_l = scan_sh[2], n_xl = scan_sh[3];
        const uint32_t rows_l = (n_l + 63u) >> 6, rows_m = (n_m + 63u) >> 6, rows_s = (n_s + 63u) >> 6, rows_all = rows_l + rows_m + rows_s;
        const bool use_mt = mt != nullptr && !(dbg & 8);
        n_mine = (n_xl + 3u - (uint32_t)wid) >> 2;
        for (uint32_t r = (uint32_t)wid;; r += 4u) {
            const bool more = r < rows_all;
            if (!more || n_mine + 64u > MLW) {
                // The listed pieces, sixty-four at a time whatever rows they came from: each claims a slot of the in-call table or finds it
                // claimed by identical bytes (by the piece's identity: the bytes themselves up to TK_XL_MAX of them; longer pieces: the first
                // and the last eight bytes, the length, and a comparison in the text).  Round 5 did this inside every row, for the 19 % of its
                // lanes whose piece was not a token -- eleven sparse passes through this code per tile where these are four.
                for (uint32_t c0 = 0; c0 < n_mine; c0 += 64u) {
#if TKF_PARK_ARGS == 2
                    mt = fresh_mt(mt_mask);  // (the in-call table's base and mask: alive inside a dense pass only)
#endif
                    const uint32_t q = c0 + (uint32_t)lane;
                    const bool have = q < n_mine;
                    const uint32_t e = mlw[have ? q : 0u];
                    const uint32_t pos = e & 4095u;
                    uint32_t k = (e >> 12) & 4095u, len = 0, ref = TKF_NONE;
                    uint64_t gs = 0;
                    bool fail = false;
                    if (have) {
                        const uint32_t e_loc = (e >> 24) == 3u ? far_end(pos) : pos + near_len(pos);
                        len = e_loc - pos;
                        gs = (uint64_t)(base + pos);
                        if (dbg & (2 | 8)) {  // (perf experiments / piece starts only: every probe counts as a hit -- only the longest pieces get here)
                            out.res[run_base + k] = (dbg & 2) ? len : 0u;
                        } else {
                            fail = true;
                            if (use_mt && len <= TK_GLANE_MAX) {
                                const bool in_lds = e_loc + 8u <= (uint32_t)TK2_WIN, exact = len <= TK_XL_MAX;
                                uint64_t w0, w1, w2;
                                tk_ident([&](uint32_t o) { return in_lds ? tk_lds_load8(raw, pos + o) : tk_load8(text, gs + o); }, len, (uint32_t)gs, w0, w1, w2);
                                unsigned long long kk = tk_ident_hash(w0, w1, w2, exact);
                                if (dbg & 512) kk &= 0xFFFull;  // test hook: force collisions between different pieces
                                if (kk == TK_EMPTY_KEY) kk = 0;
                                const uint32_t i = ((uint32_t)kk ^ (uint32_t)(kk >> 40)) & mt_mask;
                                ref = claim(w0, w1, w2, kk, exact, in_lds, pos, gs, len, i, *(const ulonglong2*)&mt[i].key, *(const ulonglong2*)&mt[i].w1);
                                fail = ref == TKF_NONE;
                            }
                        }
                    }
                    put_ref(ref != TKF_NONE, k, ref);
                    if (__ballot(fail)) {
                        // What the in-call table could not take (next to nothing: chunks too small for a table, a full neighbourhood, pieces of more
                        // than TK_GLANE_MAX bytes): an overflow entry behind the table's (one returning atomic per wavefront that has such pieces).
                        bool over = fail;
                        ref = TKF_NONE;
                        if (over && len > TK_XL_MAX && len <= T.max_token_len && (len > TK_GLANE_MAX || !mt)) {
                            // (nobody else will ask whether it is a token: tk_k_bincount looks at table slots and overflow entries of at most
                            // TK_GLANE_MAX bytes -- the latter is asked twice then, harmlessly)
                            const uint32_t rk = tk_lookup_text_piece(T, text, gs, len);
                            if (rk != TK_RANK_MAX) {
                                out.res[run_base + k] = rk;
                                over = false;  // (settled: a token after all)
                            }
                        }
                        const uint64_t m = __ballot(over);
                        if (m) {
#if TKF_PARK_ARGS == 2
                            const TkFrontOut out = fresh_out();  // (shadows the view that holds the result words' base only)
#endif
                            const int leader = __ffsll((unsigned long long)m) - 1;
                            uint32_t at = 0;
                            if (lane == leader) at = atomicAdd(&out.counters[TK_CNT_OVF], (uint32_t)__popcll(m));
                            at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            if (over && at < out.ovf_cap) {  // (beyond the capacity: the counter tells the host, which repeats the batch with more room)
                                ref = out.data.ovf_base + at;
                                *(uint4*)&out.data.ovf[at].start = make_uint4((uint32_t)gs, len, 0u, 0u);
                                if (len > TK_GLANE_MAX) tk_append_tree(out.listC, out.counters, ref, (uint32_t)gs, len);
                            }
                        }
                        put_ref(over, k, ref);
                    }
                }
                n_mine = 0;
                TKT(9);
            }
            if (!more) break;
            bool miss = false;
            uint32_t e = 0;
            if (r < rows_l) {
                const TkTables Tr = T;  // (TKF_PARK_ARGS == 2: this row's table, loaded here -- on its way while the row reads its list)
                const uint32_t q = r * 64u + (uint32_t)lane;
                if (q < n_l) {
                    const uint32_t pos = ord_sl[1023u - q], k = ordk_sl[1023u - q];
                    const uint32_t len = near_len(pos);
                    const bool in_lds = pos + len + 8u <= (uint32_t)TK2_WIN;
                    uint32_t rk = len;
                    if (!(dbg & 2)) {
                        const uint64_t gs = (uint64_t)(base + pos);
                        uint64_t w0, w1, w2;
                        tk_ident([&](uint32_t o) { return in_lds ? tk_lds_load8(raw, pos + o) : tk_load8(text, gs + o); }, len, 0u, w0, w1, w2);
                        rk = tk_probe_xl(Tr, w0, w1, w2);
                    }
                    if (rk != TK_RANK_MAX || (dbg & 8)) out.res[run_base + k] = rk == TK_RANK_MAX ? 0u : rk;
                    else {
                        miss = true;
                        e = pos | (k << 12) | (2u << 24);
                    }
                }
            } else if (r < rows_l + rows_m) {
                const TkTables Tr = T;
                const uint32_t q = (r - rows_l) * 64u + (uint32_t)lane;
                if (q < n_m) {
                    const uint32_t pos = ord_m[q], k = ordk_m[q];
                    const uint32_t len = near_len(pos);
                    const uint64_t key_m = tk_mask_low_bytes(tk_lds_load8(raw, pos), len);
                    const uint32_t rk = (dbg & 2) ? len : tk_probe_mid(Tr, key_m, len);
                    if (rk != TK_RANK_MAX || (dbg & 8)) out.res[run_base + k] = rk == TK_RANK_MAX ? 0u : rk;
                    else {
                        miss = true;
                        e = pos | (k << 12) | (1u << 24);
                    }
                }
            } else {
                const TkTables Tr = T;
                const uint32_t q = (r - rows_l - rows_m) * 64u + (uint32_t)lane;
                if (q < n_s) {
                    const uint32_t pos = ord_sl[q], k = ordk_sl[q];
                    const uint32_t len = near_len(pos);
                    const uint32_t v = __builtin_amdgcn_alignbyte(dwr[(pos >> 2) + 1], dwr[pos >> 2], pos & 3u);
                    const uint32_t key_s = v & (0xFFFFFFFFu >> (32u - 8u * len));
                    const uint32_t rk = (dbg & 2) ? len : (short_tab ? tk_probe_short(Tr, key_s, len) : tk_probe_mid(Tr, (uint64_t)key_s, len));
                    if (rk != TK_RANK_MAX || (dbg & 8)) out.res[run_base + k] = rk == TK_RANK_MAX ? 0u : rk;
                    else {
                        miss = true;
                        e = pos | (k << 12);
                    }
                }
            }
            const uint64_t mm = __ballot(miss);
            if (miss) mlw[n_mine + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = e;
            n_mine += (uint32_t)__popcll(mm);
            TKT(10);
        }
        TKT(7);
        __syncthreads();  // (the lists are reused by the next part; the counts of the entries at the tail are read below)
        TKT(8);
    }
    if (tid == 0 && np) {
        out.res[run_base + TKF_TAIL_NMISS] = ntail_sh;
        out.res[run_base + TKF_TAIL_NGAP] = GEN ? ngap_sh : 0u;
    }
    }  // (!SLOW)
#if TKF_PARK_ARGS == 2
#undef T
#if TKF_FRESH_SCALARS
#undef dbg
#endif
#undef TKF_KA_OP
#undef TKF_KA_T32
#undef TKF_KA_TP
#undef TKF_KA32
#undef TKF_KA64
#endif
#if TKF_FRESH_SCALARS
#undef tile_start
#undef tile_end
#undef base
#endif
    } while (PERSIST && next_item());
}

// The kernarg offsets TKF_PARK_ARGS == 2 relies on, checked on the device (tk_create runs this once with arguments of distinct values): the same
// parameter list as tk_k_front, so the same kernarg layout.  ok[0] = 1 when every field read through TkFrontArgs' offsets equals the argument.
__global__ void tk_k_front_args_check(TkTables T, const uint8_t* text, uint64_t n, uint64_t chunk_base, const uint32_t* brk, const uint32_t* docb, const uint32_t* ss,
                                      const uint32_t* si, TkFrontOut out, TkMissKey* mt, uint32_t mt_mask, uint32_t* deferred, const uint32_t* gapb, int dbg, uint32_t* ok) {
    typedef const __attribute__((address_space(4))) uint8_t* KArg;
    KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    auto q = [&](size_t off) { return *(const __attribute__((address_space(4))) uint64_t*)(ka + off); };
    auto d = [&](size_t off) { return *(const __attribute__((address_space(4))) uint32_t*)(ka + off); };
    const size_t t0 = offsetof(TkFrontArgs, T), o0 = offsetof(TkFrontArgs, out);
    bool good = q(t0 + offsetof(TkTables, short_tab)) == (uint64_t)(uintptr_t)T.short_tab && d(t0 + offsetof(TkTables, short_mask)) == T.short_mask &&
                d(t0 + offsetof(TkTables, short_shift)) == T.short_shift && q(t0 + offsetof(TkTables, mid_tab)) == (uint64_t)(uintptr_t)T.mid_tab &&
                d(t0 + offsetof(TkTables, mid_mask)) == T.mid_mask && d(t0 + offsetof(TkTables, mid_shift)) == T.mid_shift &&
                q(t0 + offsetof(TkTables, xl)) == (uint64_t)(uintptr_t)T.xl && d(t0 + offsetof(TkTables, xl_mask)) == T.xl_mask &&
                d(t0 + offsetof(TkTables, max_token_len)) == T.max_token_len && q(t0 + offsetof(TkTables, tok_bytes)) == (uint64_t)(uintptr_t)T.tok_bytes &&
                q(t0 + offsetof(TkTables, piece)) == (uint64_t)(uintptr_t)T.piece && q(t0 + offsetof(TkTables, piece_off)) == (uint64_t)(uintptr_t)T.piece_off &&
                q(t0 + offsetof(TkTables, piece_mask)) == T.piece_mask && d(t0 + offsetof(TkTables, n_spec)) == T.n_spec &&
                q(t0 + offsetof(TkTables, spec_bytes)) == (uint64_t)(uintptr_t)T.spec_bytes && q(t0 + offsetof(TkTables, spec_id)) == (uint64_t)(uintptr_t)T.spec_id &&
                q(t0 + offsetof(TkTables, spec_off)) == (uint64_t)(uintptr_t)T.spec_off;
    good = good && q(offsetof(TkFrontArgs, text)) == (uint64_t)(uintptr_t)text && q(offsetof(TkFrontArgs, n)) == n && q(offsetof(TkFrontArgs, chunk_base)) == chunk_base &&
           q(offsetof(TkFrontArgs, brk)) == (uint64_t)(uintptr_t)brk && q(offsetof(TkFrontArgs, docb)) == (uint64_t)(uintptr_t)docb &&
           q(offsetof(TkFrontArgs, ss)) == (uint64_t)(uintptr_t)ss && q(offsetof(TkFrontArgs, si)) == (uint64_t)(uintptr_t)si &&
           q(offsetof(TkFrontArgs, mt)) == (uint64_t)(uintptr_t)mt && d(offsetof(TkFrontArgs, mt_mask)) == mt_mask &&
           q(offsetof(TkFrontArgs, deferred)) == (uint64_t)(uintptr_t)deferred && q(offsetof(TkFrontArgs, gapb)) == (uint64_t)(uintptr_t)gapb &&
           d(offsetof(TkFrontArgs, dbg)) == (uint32_t)dbg;
    good = good && q(o0 + offsetof(TkFrontOut, starts)) == (uint64_t)(uintptr_t)out.starts && q(o0 + offsetof(TkFrontOut, tile_np)) == (uint64_t)(uintptr_t)out.tile_np &&
           q(o0 + offsetof(TkFrontOut, res)) == (uint64_t)(uintptr_t)out.res && q(o0 + offsetof(TkFrontOut, tile_sum)) == (uint64_t)(uintptr_t)out.tile_sum &&
           q(o0 + offsetof(TkFrontOut, data) + offsetof(TkMiss, tab)) == (uint64_t)(uintptr_t)out.data.tab &&
           q(o0 + offsetof(TkFrontOut, data) + offsetof(TkMiss, ovf)) == (uint64_t)(uintptr_t)out.data.ovf &&
           d(o0 + offsetof(TkFrontOut, data) + offsetof(TkMiss, ovf_base)) == out.data.ovf_base && d(o0 + offsetof(TkFrontOut, ovf_cap)) == out.ovf_cap &&
           q(o0 + offsetof(TkFrontOut, listC)) == (uint64_t)(uintptr_t)out.listC && q(o0 + offsetof(TkFrontOut, counters)) == (uint64_t)(uintptr_t)out.counters;
    if (threadIdx.x == 0 && blockIdx.x == 0) ok[0] = good ? 1u : 0u;
}

// The distinct missed pieces -- the claimed slots of the miss table and the overflow entries behind them (TkMissData) -- have to be
// listed by length bin for the merge kernels.  Same-address returning atomics run at only 25..130 M/s on this multi-XCD part (one per
// (wave, bin) cost 4.5 ms per GiB), so the lists are built without any: pass 1 (tk_k_bincount) counts per (wave, bin);
// tk_k_scan_small turns the counts into offsets -- bin-major, so that the bins lie back to back in ONE list sized by the entries, not by
// the worst case of every bin; pass 2 (tk_k_binfill) walks the same entries in the same order and writes their indices.  Both passes
// use the same wave -> range mapping: the entries that exist, in equal shares.
#define TKD_WAVES 8192  // most waves of the two passes (2048 workgroups); small inputs launch fewer

// entries of the miss data that exist: the table's slots and the overflow entries handed out
__device__ __forceinline__ uint32_t tk_miss_entries(const uint32_t* __restrict__ counters, uint32_t ovf_base, uint32_t ovf_cap) {
    const uint32_t no = counters[TK_CNT_OVF];
    return ovf_base + (no < ovf_cap ? no : ovf_cap);
}
// The length bin of an entry (TK_NBIN: none -- a free slot, or a piece of more than TK_GLANE_MAX bytes, which is on the tree list)
// (a table slot is read from the KEY table -- the last word of its 32 bytes: the claimant's identity holds the length -- so that the walk
// over four million slots does not touch the 64-byte entries)
// Round 5: the front kernel sends every piece of more than TK_XL_MAX bytes here without a look at the vocabulary (2 % of the pieces, next to
// none of them tokens).  Whether such a piece IS a token (src/lib.rs:367) is asked here, once per distinct piece: then its entry gets the
// one token and the merge kernels never see it.
// length (0: a free slot / no entry) and, for a piece of more than TK_XL_MAX bytes, start of entry i
__device__ __forceinline__ uint32_t tk_miss_len(const TkMiss& data, const TkMissKey* __restrict__ mt, uint32_t i, uint32_t hi, uint32_t& start) {
    start = 0;
    if (i >= hi) return 0u;
    if (i < data.ovf_base) {
        const unsigned long long w2 = mt[i].w2;  // (written by whoever claimed the slot; ~0: nobody did -- or tk_k_bincount has found the piece to be a token)
        if (w2 == TK_EMPTY_KEY) return 0u;
        start = (uint32_t)w2;
        return (w2 >> 63) ? ((uint32_t)(w2 >> 32) & 0x7FFFFFFFu) : (uint32_t)(w2 >> 56);
    }
    const uint2 sl = *(const uint2*)&data.ovf[i - data.ovf_base].start;
    start = sl.x;
    return sl.y;
}
__device__ __forceinline__ bool tk_miss_may_be_token(const TkTables& T, uint32_t len) { return len > TK_XL_MAX && len <= T.max_token_len && len <= TK_GLANE_MAX; }
__device__ __forceinline__ uint32_t tk_miss_bin_of_len(uint32_t len) { return (len == 0u || len > TK_GLANE_MAX) ? (uint32_t)TK_NBIN : (uint32_t)tk_bin_of(len); }

// pass 1, and the look-up of the long pieces: a wavefront takes 256 entries at a time (four loads per lane in flight), a lane then looks its
// candidates up one after the other (a wavefront has one or two among 256 slots); a piece that IS a token gets the token as its result and
// its slot's identity word is wiped, so that pass 2 takes the slot for a free one (overflow entries are looked up again there: rare).
// Before a table slot's piece is looked up, the filter of the long tokens is asked with the hash the slot was claimed under (use_filter = 0:
// the test hook that truncates those hashes is on): one in sixty of the candidates passes.
__global__ __launch_bounds__(256) void tk_k_bincount(TkTables T, const uint8_t* __restrict__ text, TkMiss data, TkMissKey* mt, uint32_t ovf_cap,
                                                     const uint32_t* __restrict__ counters, uint32_t* __restrict__ wbin, int use_filter) {
    const uint32_t ovf_base = data.ovf_base;
    const uint32_t nwaves = gridDim.x * 4u;  // tk_k_binfill runs with the same grid: identical wave -> range mapping
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t E = tk_miss_entries(counters, ovf_base, ovf_cap);
    const uint32_t per = ((E + nwaves - 1u) / nwaves + 63u) & ~63u;
    const uint64_t lo64 = (uint64_t)wave * per;
    const uint32_t lo = lo64 < E ? (uint32_t)lo64 : E, hi = lo64 + per < E ? (uint32_t)(lo64 + per) : E;
    uint32_t nb[TK_NBIN];
#pragma unroll
    for (int b = 0; b < TK_NBIN; ++b) nb[b] = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 256) {
        uint32_t len[4], start[4], cand = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + (uint32_t)j * 64u + (uint32_t)lane;
            len[j] = tk_miss_len(data, mt, i, hi, start[j]);
            if (tk_miss_may_be_token(T, len[j])) {
                bool pass = true;
                if (use_filter && i < ovf_base) {
                    const uint32_t bit = tk_xfilter_bit(mt[i].key);
                    pass = (T.xfilter[bit >> 5] >> (bit & 31u)) & 1u;
                }
                if (pass) cand |= 1u << j;
            }
        }
        while (__ballot(cand != 0u)) {
            if (cand) {
                const uint32_t j = (uint32_t)__ffs((int)cand) - 1u;
                cand &= cand - 1u;
                const uint32_t st = j == 0u ? start[0] : (j == 1u ? start[1] : (j == 2u ? start[2] : start[3]));
                const uint32_t ln = j == 0u ? len[0] : (j == 1u ? len[1] : (j == 2u ? len[2] : len[3]));
                const uint32_t rk = tk_lookup_text_piece(T, text, st, ln);
                if (rk != TK_RANK_MAX) {
                    const uint32_t i = i0 + j * 64u + (uint32_t)lane;
                    data.put(i, 1u, rk);
                    if (i < ovf_base) mt[i].w2 = TK_EMPTY_KEY;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (j == (uint32_t)jj) len[jj] = 0u;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t bin = tk_miss_bin_of_len(len[j]);
#pragma unroll
            for (int b = 0; b < TK_NBIN; ++b) nb[b] += (uint32_t)__popcll(__ballot(bin == (uint32_t)b));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < TK_NBIN; ++b) wbin[(uint32_t)b * nwaves + wave] = nb[b];  // bin-major: one scan gives every offset
    }
}

// pass 2: wscan = exclusive scan of wbin (TK_NBIN * nwaves + 1 entries; the last one is the grand total).  listB[wscan[b * nwaves]...]
// is bin b's list of data indices; its start and length go to the counters (TK_CNT_BOFF0 + b, TK_CNT_BIN0 + b) for the merge kernels.
__global__ __launch_bounds__(256) void tk_k_binfill(TkTables T, const uint8_t* __restrict__ text, TkMiss data, const TkMissKey* __restrict__ mt, uint32_t ovf_cap,
                                                    const uint32_t* __restrict__ wscan, uint32_t* __restrict__ listB, uint32_t* __restrict__ counters) {
    const uint32_t ovf_base = data.ovf_base;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t nwaves = gridDim.x * 4u;
    const uint32_t E = tk_miss_entries(counters, ovf_base, ovf_cap);
    const uint32_t per = ((E + nwaves - 1u) / nwaves + 63u) & ~63u;
    const uint64_t lo64 = (uint64_t)wave * per;
    const uint32_t lo = lo64 < E ? (uint32_t)lo64 : E, hi = lo64 + per < E ? (uint32_t)(lo64 + per) : E;
    uint32_t at[TK_NBIN];
#pragma unroll
    for (int b = 0; b < TK_NBIN; ++b) at[b] = wscan[(uint32_t)b * nwaves + wave];
    if (wave == 0 && lane < TK_NBIN) {
        counters[TK_CNT_BIN0 + lane] = wscan[(uint32_t)(lane + 1) * nwaves] - wscan[(uint32_t)lane * nwaves];
        counters[TK_CNT_BOFF0 + lane] = wscan[(uint32_t)lane * nwaves];
    }
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        uint32_t start, len = tk_miss_len(data, mt, i, hi, start);
        if (i >= ovf_base && tk_miss_may_be_token(T, len) && tk_lookup_text_piece(T, text, start, len) != TK_RANK_MAX) len = 0u;  // (pass 1 has given it its token)
        const uint32_t bin = tk_miss_bin_of_len(len);
#pragma unroll
        for (int b = 0; b < TK_NBIN; ++b) {
            const uint64_t m = __ballot(bin == (uint32_t)b);
            if (bin == (uint32_t)b) listB[at[b] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
            at[b] += (uint32_t)__popcll(m);
        }
    }
}

// ------------------------------------------------------------------------------------------
// merges on lists of miss-data indices: byte_pair_merge (src/lib.rs:140-196) of the piece text[start .. start + len) of entry i
// result: data[i].res = {token count, the token (count 1) | the staging position of the tokens}
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void tk_put_result(const TkMiss& data, uint32_t i, uint32_t cnt, uint32_t tok) { data.put(i, cnt, tok); }
template <int NMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void tk_k_merge_llane(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ listB,
                                                             const uint32_t* __restrict__ counters, int bin, TkMiss data,
                                                             uint32_t* __restrict__ staging) {
    const uint32_t count = counters[TK_CNT_BIN0 + bin];  // list start and length produced on the device (tk_k_binfill): no host round trip before the merges
    const uint32_t* __restrict__ list = listB + counters[TK_CNT_BOFF0 + bin];
    __shared__ uint32_t s_id[NMAX * THREADS];
    __shared__ uint32_t s_rk[NMAX * THREADS];
    uint32_t* id = s_id + threadIdx.x;
    uint32_t* rk = s_rk + threadIdx.x;
    for (uint32_t it = blockIdx.x * THREADS + threadIdx.x; it < count; it += gridDim.x * THREADS) {
        const uint32_t mi = list[it];
        const uint2 pc = data.piece(mi);
        const uint32_t s = pc.x, n = pc.y;
        const uint32_t t = tk_lane_merge<THREADS>(T, text, s, n, id, rk, staging + s);
        tk_put_result(data, mi, t, t == 1 ? id[0] : s);
    }
}

template <int G>
__global__ __launch_bounds__(256) void tk_k_merge_group(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ listB,
                                                         const uint32_t* __restrict__ counters, int bin, TkMiss data,
                                                         uint32_t* __restrict__ staging) {
    const uint32_t count = counters[TK_CNT_BIN0 + bin];
    const uint32_t* __restrict__ list = listB + counters[TK_CNT_BOFF0 + bin];
    constexpr int C = 16, NMAX = G * C, PPW = 64 / G;
    constexpr uint32_t NONE = 0xFFFFu;
    __shared__ __attribute__((aligned(16))) uint32_t s_id[4][1024];
    __shared__ __attribute__((aligned(16))) uint32_t s_rk[4][1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane & (G - 1), grp = lane / G, gbase = grp * G;
    uint32_t* id = s_id[wid] + grp * NMAX;
    uint32_t* rk = s_rk[wid] + grp * NMAX;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    // lowest rank of the lane's 16 positions and the leftmost position that has it (32-bit compares: a 64-bit (rank, position) key
    // costs three vector instructions per compare, and this runs after every merge)
    auto local_min = [&](uint32_t& pos_out) -> uint32_t {
        const uint4* q = (const uint4*)(rk + g * C);
        uint32_t r[C];
#pragma unroll
        for (int v = 0; v < C / 4; ++v) {
            const uint4 x = q[v];
            r[4 * v] = x.x; r[4 * v + 1] = x.y; r[4 * v + 2] = x.z; r[4 * v + 3] = x.w;
        }
        uint32_t m = r[0];
#pragma unroll
        for (int c = 1; c < C; ++c) m = r[c] < m ? r[c] : m;
        uint32_t eq = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) eq |= (uint32_t)(r[c] == m) << c;
        pos_out = (uint32_t)(g * C) + (uint32_t)__ffs((int)eq) - 1u;
        return m;
    };
    for (uint32_t e0 = wave * PPW; e0 < count; e0 += nwaves * PPW) {
        const uint32_t e = e0 + grp;
        const bool valid = e < count;
        uint32_t mi = 0, s = 0, n = 0;
        if (valid) {
            mi = list[e];
            const uint2 pc = data.piece(mi);
            s = pc.x;
            n = pc.y;
        }
        uint32_t mask = 0;
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const uint32_t k = g * C + c;
            uint32_t r = TK_RANK_MAX;
            if (k < n) {
                const uint32_t b0 = text[s + k];
                id[k] = T.byte_rank[b0];
                if (k + 1 < n) r = T.pair2[(b0 << 8) | text[s + k + 1]];
                mask |= 1u << c;
            }
            rk[k] = r;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t lpos = 0, lrank = local_min(lpos);
        for (;;) {
            // the group's lowest rank, then the leftmost position among the lanes that hold it (leftmost lowest: lib.rs:151,190)
            uint32_t best = lrank;
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                const uint32_t w = (uint32_t)__shfl_xor((int)best, o, 64);
                best = w < best ? w : best;
            }
            const bool fin = best == TK_RANK_MAX;
            if (__all(fin)) break;
            uint32_t bpos = lrank == best ? lpos : 0xFFFFFFFFu;
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                const uint32_t w = (uint32_t)__shfl_xor((int)bpos, o, 64);
                bpos = w < bpos ? w : bpos;
            }
            const uint32_t bi = bpos & (NMAX - 1), ob = bi / C, bl = bi % C;
            const uint64_t nbw = __ballot(mask != 0);
            const uint64_t nb = G == 64 ? nbw : ((nbw >> gbase) & ((1ull << (G & 63)) - 1ull));
            const uint32_t my_first = mask ? (uint32_t)(g * C + __ffs((int)mask) - 1) : NONE;
            const uint32_t my_last = mask ? (uint32_t)(g * C + 31 - __clz((int)mask)) : NONE;
            const uint32_t om = __shfl(mask, gbase + (int)ob, 64);
            uint32_t j;
            {
                const uint32_t hi = om & ~((2u << bl) - 1u);
                const uint64_t la = nb & ~((2ull << ob) - 1ull);
                const int lj = la ? __ffsll((unsigned long long)la) - 1 : 0;
                const uint32_t fj = __shfl(my_first, gbase + lj, 64);
                j = hi ? ob * C + (uint32_t)__ffs((int)hi) - 1u : fj;
            }
            j &= (NMAX - 1);
            const uint32_t oj = j / C, jl = j % C;
            uint32_t nn;
            {
                const uint32_t ojm = __shfl(mask, gbase + (int)oj, 64);
                const uint32_t hi = ojm & ~((2u << jl) - 1u);
                const uint64_t la = nb & ~((2ull << oj) - 1ull);
                const int ln = la ? __ffsll((unsigned long long)la) - 1 : 0;
                const uint32_t fn = __shfl(my_first, gbase + ln, 64);
                nn = hi ? oj * C + (uint32_t)__ffs((int)hi) - 1u : (la ? fn : NONE);
            }
            uint32_t pp;
            {
                const uint32_t lo = om & ((1u << bl) - 1u);
                const uint64_t lb = nb & ((1ull << ob) - 1ull);
                const int lp = lb ? 63 - __clzll((long long)lb) : 0;
                const uint32_t fl = __shfl(my_last, gbase + lp, 64);
                pp = lo ? ob * C + 31u - (uint32_t)__clz((int)lo) : (lb ? fl : NONE);
            }
            uint32_t newr = TK_RANK_MAX;
            if (!fin) {
                if (g == 0 && nn != NONE) newr = tk_probe_pair(T, best, id[nn]);
                if (g == 1 && pp != NONE) newr = tk_probe_pair(T, id[pp], best);
            }
            const uint32_t newr_i = __shfl(newr, gbase, 64), newr_p = __shfl(newr, gbase + 1, 64);
            __builtin_amdgcn_wave_barrier();
            bool touched = false;
            if (!fin) {
                if (g == (int)ob) {
                    id[bi] = best;
                    rk[bi] = newr_i;
                    touched = true;
                }
                if (g == (int)oj) {
                    mask &= ~(1u << jl);
                    rk[j] = TK_RANK_MAX;
                    touched = true;
                }
                if (pp != NONE && g == (int)(pp / C)) {
                    rk[pp] = newr_p;
                    touched = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (touched) lrank = local_min(lpos);
        }
        const uint32_t mine = __popc(mask);
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            uint32_t w = __shfl_up(inc, o, 64);
            if (g >= o) inc += w;
        }
        const uint32_t total = __shfl(inc, gbase + G - 1, 64);
        if (valid) {
            uint32_t t = inc - mine, mm = mask;
            while (mm) {
                const int c = __ffs((int)mm) - 1;
                mm &= mm - 1;
                staging[s + t++] = id[g * C + c];
            }
            if (g == 0) tk_put_result(data, mi, total, total == 1 ? id[0] : s);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// All the binned pieces (2 .. TK_GLANE_MAX bytes) in ONE launch.  The nine kernels above -- one per length bin, on side streams -- ran
// as three chains on the device's hardware queues, each kernel as long as its slowest wavefront: 1.9 ms per GiB, of which the three
// one-lane-per-piece kernels for 25..64-byte pieces took 0.25..0.73 ms each (a wavefront alone on its SIMD, a rank scan of up to 63 LDS
// reads per merge).  Here a wavefront works on `units`: 64 / G pieces of one bin, G = 1 .. 64 lanes per piece, 16 part positions per
// lane (the scheme of tk_k_merge_group with G a run-time value of the unit), taken longest bin first -- the first unit by the
// wavefront's index, the next ones through sixteen work counters -- so the long chains of merges start at once and the short
// pieces fill in behind them.
//   * a position's LDS word is the KEY rank << 10 | position (packed pair table: ranks < 2^22), TKM_NOKEY for "no pair starts here":
//     the lowest key is the leftmost lowest rank (lib.rs:151,190) -- one min-reduction per merge instead of two, and a lane's minimum over
//     its 16 positions is four 16-byte LDS reads and fifteen v_min;
//   * G = 1 (pieces of <= 16 bytes): no cross-lane step at all; the lane issues the first buckets of its two probes together.
// Needs the packed pair table (every id <= TK_PAIR8_MAX_ID); vocabularies with larger ids keep the kernels above.
// ------------------------------------------------------------------------------------------
#define TKM_NOKEY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t tkm_key(uint32_t rank, uint32_t pos) { return rank == TK_RANK_MAX ? TKM_NOKEY : ((rank << 10) | pos); }
__device__ __forceinline__ int tkm_lg_of_bin(int b) {  // log2 of the lanes per piece: 16 positions per lane
    return b == 0 ? 0 : (b <= 2 ? 1 : (b <= 4 ? 2 : b - 2));  // bins <= 16, 24, 32, 48, 64, 128, 256, 512, 1024 bytes
}
// minimum over the aligned group of 1 << lg lanes, in every lane of it: DPP steps inside a row of 16 lanes (a few cycles each; a
// bpermute goes through the LDS crossbar and this sits on the dependent chain of every merge), bpermutes only for 32 and 64 lanes
__device__ __forceinline__ uint32_t tkm_group_min(uint32_t v, int lg) {
    if (lg >= 1) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]: lane ^ 1
    if (lg >= 2) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]: lane ^ 2
    if (lg >= 3) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror: the other quad of 8
    if (lg >= 4) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror: the other half of 16
    if (lg >= 5) {
        // (32 and 64 lanes: the rows' minima through the scalar unit -- four v_readlane and three s_min -- instead of two dependent bpermutes
        // through the LDS crossbar, on the chain of every merge of the longest pieces)
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
        const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
        const uint32_t lo = min(r0, r1), hi = min(r2, r3);
        v = lg >= 6 ? min(lo, hi) : ((threadIdx.x & 32u) ? hi : lo);
    }
    return v;
}
// both probes of a merge by one lane: the first buckets of the two in flight together (packed table)
__device__ __forceinline__ void tkm_probe2(const TkTables& T, uint32_t a0, uint32_t b0, bool on0, uint32_t a1, uint32_t b1, bool on1, uint32_t& r0, uint32_t& r1) {
    r0 = r1 = TK_RANK_MAX;
    const uint64_t k0 = ((uint64_t)a0 << TK_PAIR8_ID_BITS) | b0, k1 = ((uint64_t)a1 << TK_PAIR8_ID_BITS) | b1;
    uint64_t bk0 = tk_pair_slot_hash(k0) & T.pair_mask, bk1 = tk_pair_slot_hash(k1) & T.pair_mask;
    while (on0 || on1) {
        ulonglong2 x0 = {0, 0}, x1 = {0, 0}, y0 = {0, 0}, y1 = {0, 0};
        if (on0) {
            x0 = *(const ulonglong2*)(T.pair8 + bk0 * 4);
            x1 = *(const ulonglong2*)(T.pair8 + bk0 * 4 + 2);
        }
        if (on1) {
            y0 = *(const ulonglong2*)(T.pair8 + bk1 * 4);
            y1 = *(const ulonglong2*)(T.pair8 + bk1 * 4 + 2);
        }
        if (on0) {
            const uint64_t s[4] = {x0.x, x0.y, x1.x, x1.y};
            on0 = s[3] != TK_EMPTY_KEY;  // (a bucket with a free slot ends the search)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((s[q] >> 22) == k0) {
                    r0 = (uint32_t)(s[q] & 0x3FFFFFu);
                    on0 = false;
                }
            bk0 = (bk0 + 1) & T.pair_mask;
        }
        if (on1) {
            const uint64_t s[4] = {y0.x, y0.y, y1.x, y1.y};
            on1 = s[3] != TK_EMPTY_KEY;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((s[q] >> 22) == k1) {
                    r1 = (uint32_t)(s[q] & 0x3FFFFFu);
                    on1 = false;
                }
            bk1 = (bk1 + 1) & T.pair_mask;
        }
    }
}
#ifndef TKM_WAVES
#define TKM_WAVES 4       // wavefronts per workgroup (8 KiB of LDS each)
#endif
#ifndef TKM_WGS_PER_CU
#define TKM_WGS_PER_CU (16 / TKM_WAVES)  // 40 KiB of LDS per workgroup of four
#endif
#define TKM_LDS_BYTES (TKM_WAVES * (2 * 1024 * 4 + 1024 * 2))
// Round 5: a part's neighbours are LINKS in LDS -- the position of the next part in the upper eleven bits of its id word (ids of the packed
// pair table have 21 bits), the previous part's in a halfword of its own -- read by all lanes of the piece at once.  Round 4 found them
// through the lanes' alive masks: a ballot and three __shfl per neighbour (bpermutes through the LDS crossbar), a dozen in a row per
// step of two merges -- more of a step's time than its table probes.
#define TKM_ID_BITS 21
#define TKM_ID_MASK ((1u << TKM_ID_BITS) - 1u)
#define TKM_NO_NEXT 0x7FFu
#define TKM_NO_PREV 0xFFFFu
#ifndef TKM_MIN_WAVES_EU
#define TKM_MIN_WAVES_EU 4  // (the second launch bound is wavefronts per SIMD in HIP)
#endif
#define TKM_WORK_STRIDE 64  // words between two work counters (256 bytes)
__global__ __launch_bounds__(64 * TKM_WAVES, TKM_MIN_WAVES_EU) void tk_k_merge_all(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ listB,
                                                                      uint32_t* __restrict__ counters, TkMiss data, uint32_t* __restrict__ staging,
                                                                      uint32_t* __restrict__ work /* 16 counters, TKM_WORK_STRIDE words apart, zero */, int dbg) {
    constexpr int C = 16;
    constexpr uint32_t NONE = 0xFFFFu;
    // (dynamic LDS, TKM_LDS_BYTES at the launch: with a static size the compiler derives the occupancy from it and lets the registers
    // grow to match -- the launch bounds are what shall limit them: a workgroup has to fit the place a front-kernel workgroup leaves)
    extern __shared__ __attribute__((aligned(16))) uint32_t tkm_lds[];
    uint32_t(*s_id)[1024] = (uint32_t(*)[1024])tkm_lds;
    uint32_t(*s_key)[1024] = s_id + TKM_WAVES;
    uint16_t(*s_prv)[1024] = (uint16_t(*)[1024])(s_key + TKM_WAVES);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t* id = s_id[wid] + lane * C;   // the lane's 16 positions (a piece's G lanes are neighbours: its positions are contiguous)
    uint32_t* key = s_key[wid] + lane * C;
    const uint32_t* idw = s_id[wid];
    uint32_t* keyw = s_key[wid];
    uint16_t* prvw = s_prv[wid];
    const uint32_t nwaves = gridDim.x * (uint32_t)TKM_WAVES;  // (a multiple of 16: see the work counters)
    // units, longest bin first (wave-uniform; the counts were left by tk_k_binfill)
    uint32_t ustart[TK_NBIN + 1];  // ustart[q]: first unit of the q-th bin in processing order (bin TK_NBIN - 1 - q)
    ustart[0] = 0;
#pragma unroll
    for (int q = 0; q < TK_NBIN; ++q) {
        const int b = TK_NBIN - 1 - q;
        const uint32_t cnt = counters[TK_CNT_BIN0 + b], sh = 6u - (uint32_t)tkm_lg_of_bin(b);
        ustart[q + 1] = ustart[q] + ((cnt + (1u << sh) - 1u) >> sh);
    }
    const uint32_t total_units = ustart[TK_NBIN];
    uint32_t u = blockIdx.x * (uint32_t)TKM_WAVES + (uint32_t)wid;
    auto local_min = [&]() -> uint32_t {
        const uint4* q = (const uint4*)key;
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        const uint32_t m0 = min(min(a.x, a.y), min(a.z, a.w)), m1 = min(min(b.x, b.y), min(b.z, b.w));
        const uint32_t m2 = min(min(c.x, c.y), min(c.z, c.w)), m3 = min(min(d.x, d.y), min(d.z, d.w));
        return min(min(m0, m1), min(m2, m3));
    };
    while (u < total_units) {
        int q = 0;
        uint32_t ubase = 0;  // = ustart[q] (no indexing by a run-time value: the table stays in scalar registers)
#pragma unroll
        for (int i = 1; i < TK_NBIN; ++i)
            if (u >= ustart[i]) {
                q = i;
                ubase = ustart[i];
            }
        const int b = TK_NBIN - 1 - q;
        const int lg = tkm_lg_of_bin(b);
        const uint32_t G = 1u << lg, NMAX = G * C, ppw = 64u >> lg;
        const uint32_t g = (uint32_t)lane & (G - 1u), grp = (uint32_t)lane >> lg, gbase = grp << lg;
        // Long pieces are the kernel's critical path (a 1 KiB piece is ~700 merges, one after the other): their wavefronts get the SIMD's
        // issue slots first, the short pieces' wavefronts fill the gaps
        if (b >= 7) __builtin_amdgcn_s_setprio(3);
        else if (b >= 5) __builtin_amdgcn_s_setprio(2);
        else if (b >= 3) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const uint32_t count = counters[TK_CNT_BIN0 + b];
        const uint32_t e = (u - ubase) * ppw + grp;
        const bool valid = e < count;
        uint32_t mi = 0, s = 0, n = 0;
        if (valid) {
            mi = listB[counters[TK_CNT_BOFF0 + b] + e];
            const uint2 pc = data.piece(mi);
            s = pc.x;
            n = pc.y;
        }
        // the lane's 16 parts: ids of the single bytes, keys of the 2-byte pairs (17 text bytes: three aligned words)
        uint32_t mask = 0;
        {
            const uint32_t k0 = g * C;
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            if (k0 < n) {
                w0 = tk_load8(text, (uint64_t)s + k0);
                w1 = tk_load8(text, (uint64_t)s + k0 + 8);
                w2 = tk_load8(text, (uint64_t)s + k0 + 16);
            }
            uint32_t rr[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const uint32_t k = k0 + c;
                const uint32_t b0 = (uint32_t)((c < 8 ? w0 >> (8 * c) : w1 >> (8 * (c - 8))) & 0xFFu);
                const uint32_t b1 = (uint32_t)((c < 7 ? w0 >> (8 * (c + 1)) : (c < 15 ? w1 >> (8 * (c - 7)) : w2)) & 0xFFu);
                // (both look-ups by loads that stand in no `if` of their own: the sixteen positions' loads are in flight together)
                const uint32_t br = T.byte_rank[k < n ? b0 : 0u], p2 = T.pair2[k + 1 < n ? ((b0 << 8) | b1) : 0u];
                rr[c] = k + 1 < n ? p2 : TK_RANK_MAX;
                if (k < n) {
                    id[c] = br | ((k + 1 < n ? k + 1u : TKM_NO_NEXT) << TKM_ID_BITS);  // (positions are relative to the piece: k = g * C + c)
                    prvw[gbase * C + k] = (uint16_t)(k ? k - 1u : TKM_NO_PREV);
                    mask |= 1u << c;
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) key[c] = tkm_key(rr[c], k0 + c);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t lkey = local_min();
        // (perf experiments, debug bits 25..28 = bin + 1: only that bin is merged, the pieces of the others come out as their single bytes)
        if (((dbg >> 25) & 15) && ((dbg >> 25) & 15) - 1 != b) lkey = TKM_NOKEY;
        for (;;) {
            // the piece's lowest key: leftmost lowest rank
            const uint32_t best = tkm_group_min(lkey, lg);
            const bool fin = best == TKM_NOKEY;
            if (__all(fin)) break;
            const uint32_t bi = best & (NMAX - 1u), brank = best >> 10, ob = bi / C, bl = bi % C;
            // neighbours through the links: j = the part being absorbed, nn = the part after it, pp = the part before bi
            const uint32_t* pid = idw + gbase * C;  // the piece's positions
            uint32_t* pkey = keyw + gbase * C;
            uint16_t* pprv = prvw + gbase * C;
            const uint32_t j = (pid[bi] >> TKM_ID_BITS) & (NMAX - 1u);  // (a key at bi means there is a part behind it)
            const uint32_t oj = j / C, jl = j % C;
            const uint32_t nn_raw = pid[j] >> TKM_ID_BITS, pp_raw = pprv[bi];
            const uint32_t nn = nn_raw == TKM_NO_NEXT ? NONE : nn_raw, pp = pp_raw == TKM_NO_PREV ? NONE : pp_raw;
            if (lg >= 2 && !(dbg & 0x80000)) {
                // TWO merges per step (pieces of four lanes and more; debug bit 0x80000: one).  The step's time is the latency of its
                // table probes, and a long piece is a chain of hundreds of steps.  What the reference merges next (lib.rs:151,190) is the
                // lowest key once more: either the lowest of the keys this merge leaves untouched -- known now -- or one of the two it
                // creates.  So the second-lowest untouched key's merge is prepared at once, its neighbours taken from the state this
                // merge leaves behind and its two probes sent along by lanes 2 and 3, and it is carried out iff both new keys of the first
                // merge come out above it: then it IS the reference's next merge.  (Keys hold the position: no ties.)
                // merge 1, the part that does not wait for a probe: the merged token's id, the absorbed part
                if (!fin) {
                    if (g == ob) {
                        id[bl] = brank | ((nn == NONE ? TKM_NO_NEXT : nn) << TKM_ID_BITS);
                        key[bl] = TKM_NOKEY;  // (until the probe is back: not a candidate for the second merge)
                    }
                    if (g == oj) {
                        mask &= ~(1u << jl);
                        key[jl] = TKM_NOKEY;
                    }
                    if (pp != NONE && g == pp / C) pkey[pp] = TKM_NOKEY;
                    if (nn != NONE && g == nn / C) pprv[nn] = (uint16_t)bi;
                }
                __builtin_amdgcn_wave_barrier();
                const bool touched1 = !fin && (g == ob || g == oj || (pp != NONE && g == pp / C));
                const uint32_t lkey2 = touched1 ? local_min() : lkey;
                const uint32_t best2 = fin ? TKM_NOKEY : tkm_group_min(lkey2, lg);
                const bool has2 = best2 != TKM_NOKEY;
                const uint32_t bi2 = best2 & (NMAX - 1u), rank2 = best2 >> 10, ob2 = bi2 / C, bl2 = bi2 % C;
                // (its neighbours from the state the first merge leaves behind: the links have been updated above)
                const uint32_t j2 = (pid[bi2] >> TKM_ID_BITS) & (NMAX - 1u);
                const uint32_t nn2_raw = pid[j2] >> TKM_ID_BITS, pp2_raw = pprv[bi2];
                const uint32_t nn2 = nn2_raw == TKM_NO_NEXT ? NONE : nn2_raw, pp2 = pp2_raw == TKM_NO_PREV ? NONE : pp2_raw;
                // (ONE probe region for the four lanes: four `if (g == ..) probe` statements run one after the other, each waiting for its own
                // loads -- the step would take four table latencies instead of one)
                uint32_t newr = TK_RANK_MAX;
                {
                    const uint32_t nb_pos = g == 0 ? nn : (g == 1 ? pp : (g == 2 ? nn2 : pp2));  // the neighbour this lane's pair is made with
                    const bool second = g >= 2, left = (g & 1u) != 0;                             // lanes 1, 3: (previous, merged)
                    const bool on = !fin && g < 4u && nb_pos != NONE && (!second || has2) && !(dbg & 0x1000000);  // (0x1000000, perf experiments: no probes)
                    const uint32_t nid = pid[nb_pos != NONE ? nb_pos : 0u] & TKM_ID_MASK, mid = second ? rank2 : brank;
                    if (on) newr = tk_probe_pair(T, left ? nid : mid, left ? mid : nid);
                }
                const uint32_t newr_i = __shfl(newr, gbase, 64), newr_p = __shfl(newr, gbase + 1, 64);
                const uint32_t newr_i2 = __shfl(newr, gbase + 2, 64), newr_p2 = __shfl(newr, gbase + 3, 64);
                const uint32_t nk_i = tkm_key(newr_i, bi), nk_p = pp != NONE ? tkm_key(newr_p, pp) : TKM_NOKEY;
                const bool take2 = !fin && has2 && best2 < nk_i && best2 < nk_p;
                __builtin_amdgcn_wave_barrier();
                bool touched = touched1;
                if (!fin) {
                    if (g == ob) key[bl] = nk_i;
                    if (pp != NONE && g == pp / C) pkey[pp] = nk_p;
                    if (take2) {  // (after the first merge's keys: a position the two share ends up with the second's)
                        const uint32_t oj2 = j2 / C, jl2 = j2 % C;
                        if (g == ob2) {
                            id[bl2] = rank2 | ((nn2 == NONE ? TKM_NO_NEXT : nn2) << TKM_ID_BITS);
                            key[bl2] = tkm_key(newr_i2, bi2);
                            touched = true;
                        }
                        if (g == oj2) {
                            mask &= ~(1u << jl2);
                            key[jl2] = TKM_NOKEY;
                            touched = true;
                        }
                        if (pp2 != NONE && g == pp2 / C) {
                            pkey[pp2] = tkm_key(newr_p2, pp2);
                            touched = true;
                        }
                        if (nn2 != NONE && g == nn2 / C) pprv[nn2] = (uint16_t)bi2;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (touched) lkey = local_min();
                continue;
            }
            // the two new pairs: (merged, next) and (previous, merged)
            uint32_t newr_i = TK_RANK_MAX, newr_p = TK_RANK_MAX;
            if (dbg & 0x1000000) {  // (perf experiments: no probes -- wrong tokens, the cost of everything else)
            } else if (lg == 0) {
                if (!fin) tkm_probe2(T, brank, nn != NONE ? (pid[nn & (NMAX - 1u)] & TKM_ID_MASK) : 0u, nn != NONE, pp != NONE ? (pid[pp & (NMAX - 1u)] & TKM_ID_MASK) : 0u, brank, pp != NONE, newr_i, newr_p);
            } else {
                uint32_t newr = TK_RANK_MAX;
                {  // (one probe region for both lanes: see above)
                    const uint32_t nb_pos = g == 0 ? nn : pp;
                    const bool on = !fin && g < 2u && nb_pos != NONE;
                    const uint32_t nid = pid[nb_pos != NONE ? nb_pos : 0u] & TKM_ID_MASK;
                    if (on) newr = tk_probe_pair(T, g ? nid : brank, g ? brank : nid);
                }
                newr_i = __shfl(newr, gbase, 64);
                newr_p = __shfl(newr, gbase + 1, 64);
            }
            __builtin_amdgcn_wave_barrier();
            bool touched = false;
            if (!fin) {
                if (g == ob) {
                    id[bl] = brank | ((nn == NONE ? TKM_NO_NEXT : nn) << TKM_ID_BITS);
                    key[bl] = tkm_key(newr_i, bi);
                    touched = true;
                }
                if (g == oj) {
                    mask &= ~(1u << jl);
                    key[jl] = TKM_NOKEY;
                    touched = true;
                }
                if (pp != NONE && g == pp / C) {
                    pkey[pp] = tkm_key(newr_p, pp);
                    touched = true;
                }
                if (nn != NONE && g == nn / C) pprv[nn] = (uint16_t)bi;
            }
            __builtin_amdgcn_wave_barrier();
            if (touched) lkey = local_min();
        }
        // the piece's tokens, in order, to the staging area at its text position
        const uint32_t mine = __popc(mask);
        uint32_t inc = mine;
        for (uint32_t o = 1; o < G; o <<= 1) {
            const uint32_t w = __shfl_up(inc, o, 64);
            if (g >= o) inc += w;
        }
        const uint32_t total = __shfl(inc, gbase + G - 1, 64);
        if (valid) {
            // (into the piece's entry when they fit: the back end then needs one access per occurrence, not a second one to the staging area)
            const bool fits = data.fits(mi, total);
            uint32_t* dst = fits ? data.tab[mi].tok : staging + s;
            uint32_t t = inc - mine, mm = mask;
            while (mm) {
                const int c = __ffs((int)mm) - 1;
                mm &= mm - 1;
                dst[t++] = id[c] & TKM_ID_MASK;
            }
            if (g == 0) {
                if (fits) data.put_count(mi, total | TKD_INLINE_BIT);
                else tk_put_result(data, mi, total, total == 1 ? (id[0] & TKM_ID_MASK) : s);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // The next unit: sixteen counters share the requests, each in a cache line of its own -- atomics on one LINE are served one after
        // the other at the memory side (with the sixteen in one line, 34 000 requests cost 1.1 ms: half the kernel).  Counter c hands out the
        // units nwaves + c, nwaves + c + 16, ...
        uint32_t nu = 0;
        {
            const uint32_t cidx = (blockIdx.x * (uint32_t)TKM_WAVES + (uint32_t)wid) & 15u;
            if (lane == 0) nu = nwaves + cidx + 16u * atomicAdd(&work[cidx * TKM_WORK_STRIDE], 1u);
        }
        u = (uint32_t)__shfl((int)nu, 0, 64);
    }
}

// (redo_only: after tk_k_merge_rounds -- only the pieces it has marked TK_MERGE_REDO)
__global__ __launch_bounds__(256) void tk_k_merge_long(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ listC,
                                                        uint32_t nC, uint32_t* __restrict__ g_id, uint32_t* __restrict__ g_rk,
                                                        uint32_t* __restrict__ g_nx, uint32_t* __restrict__ g_pv, uint64_t* __restrict__ g_lv,
                                                        TkMiss data, uint32_t* __restrict__ staging, int redo_only) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (uint32_t w = wave; w < nC; w += nwaves) {
        const uint32_t* ent = listC + 5 * (uint64_t)w;
        const uint32_t mi = ent[0], s = ent[1], n = ent[2];
        if (redo_only && data.result(mi).x != TK_MERGE_REDO) continue;
        uint32_t* id = g_id + ent[3];
        uint32_t* rk = g_rk + ent[3];
        uint32_t* nx = g_nx + ent[3];
        uint32_t* pv = g_pv + ent[3];
        uint64_t* lv = g_lv + ent[4];
        uint32_t cntl[TK_MAX_LEVELS + 1], offl[TK_MAX_LEVELS + 1];
        int nl = 0;
        cntl[0] = n;
        offl[0] = 0;
        {
            uint32_t c = n, o = 0;
            do {
                c = (c + 63) >> 6;
                ++nl;
                cntl[nl] = c;
                offl[nl] = o;
                o += c;
            } while (c > 64);
        }
        for (uint32_t k = lane; k < n; k += 64) {
            uint32_t b0 = text[s + k];
            id[k] = T.byte_rank[b0];
            rk[k] = k + 1 < n ? T.pair2[(b0 << 8) | text[s + k + 1]] : TK_RANK_MAX;
            nx[k] = k + 1;
            pv[k] = k - 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int l = 1; l <= nl; ++l) {
            for (uint32_t b = 0; b < cntl[l]; ++b) {
                uint32_t k = b * 64 + lane;
                uint64_t key = ~0ull;
                if (k < cntl[l - 1]) key = l == 1 ? (((uint64_t)rk[k] << 32) | k) : lv[offl[l - 1] + k];
                key = tk_wave_min_u64(key);
                if (lane == 0) lv[offl[l] + b] = key;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        for (;;) {
            uint64_t top = (uint32_t)lane < cntl[nl] ? lv[offl[nl] + lane] : ~0ull;
            top = tk_wave_min_u64(top);
            uint32_t m = (uint32_t)(top >> 32);
            if (m == TK_RANK_MAX) break;
            const uint32_t i = (uint32_t)top;
            const uint32_t j = nx[i];
            const uint32_t nn = nx[j];
            const uint32_t pp = pv[i];
            uint32_t newr = TK_RANK_MAX;
            if (lane == 0 && nn < n) newr = tk_probe_pair(T, m, id[nn]);
            if (lane == 1 && pp != 0xFFFFFFFFu) newr = tk_probe_pair(T, id[pp], m);
            if (lane == 0) {
                id[i] = m;
                nx[i] = nn;
                if (nn < n) pv[nn] = i;
                rk[j] = TK_RANK_MAX;
                id[j] = TK_RANK_MAX;  // absorbed (no token has this id): the emit below compacts on it
                rk[i] = newr;
            }
            if (lane == 1 && pp != 0xFFFFFFFFu) rk[pp] = newr;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            uint32_t bi = i, bj = j, bp = pp != 0xFFFFFFFFu ? pp : i;
            for (int l = 1; l <= nl; ++l) {
                bi >>= 6;
                bj >>= 6;
                bp >>= 6;
                for (int t = 0; t < 3; ++t) {
                    uint32_t b = t == 0 ? bp : (t == 1 ? bi : bj);
                    if ((t == 1 && bi == bp) || (t == 2 && (bj == bi || bj == bp))) continue;
                    uint32_t k = b * 64 + lane;
                    uint64_t key = ~0ull;
                    if (k < cntl[l - 1]) key = l == 1 ? (((uint64_t)rk[k] << 32) | k) : lv[offl[l - 1] + k];
                    key = tk_wave_min_u64(key);
                    if (lane == 0) lv[offl[l] + b] = key;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
        // emit the surviving parts in order: wave-wide compaction, 64 positions per step
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        uint32_t t = 0;
        for (uint32_t k0 = 0; k0 < n; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t v = k < n ? id[k] : (uint32_t)TK_RANK_MAX;
            const uint64_t m = __ballot(v != TK_RANK_MAX);
            if (v != TK_RANK_MAX) staging[s + t + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
            t += (uint32_t)__popcll(m);
        }
        if (lane == 0) tk_put_result(data, mi, t, t == 1 ? id[0] : s);  // (the leftmost part always survives)
    }
}

// ------------------------------------------------------------------------------------------
// Long pieces, in rounds (one workgroup per piece).  byte_pair_merge (src/lib.rs:140-196) always merges the leftmost pair of the lowest
// rank m.  As long as every pair that such a merge creates ranks ABOVE m -- the normal case: a longer token is learnt after its parts --
// the reference's next picks are exactly the other rank-m pairs, left to right, skipping the ones a pick has consumed.  One round therefore
// applies ALL of them at once: within a run of overlapping rank-m pairs the even ones.  The number of rounds is bounded by the number of
// distinct ranks that get merged, not by the length: a megabyte of one character takes a few dozen rounds.
// The condition is CHECKED, round by round, on every pair a merge creates -- the lasting ones and the ones that exist only between two
// picks of the same round (merged part, not-yet-merged right neighbour).  A piece that violates it is handed to tk_k_merge_long (one merge
// at a time, the reference's order literally), so the result never depends on the assumption.
//   P0/R0 and P1/R1: the parts (token ids) and the rank of each part's pair with its right neighbour, double-buffered and compacted every round.
// ------------------------------------------------------------------------------------------
#define TKB_THREADS 1024
struct TkRunState {  // segment summary for the run-parity scan: is the segment all rank-m pairs, parity of its trailing run of them
    uint32_t all, par;
};
__device__ __forceinline__ TkRunState tk_run_combine(TkRunState a, TkRunState b) { return TkRunState{a.all & b.all, b.all ? (a.par ^ b.par) : b.par}; }

// inclusive scan of run states over a wavefront (lane order = element order)
__device__ __forceinline__ TkRunState tk_run_scan_wave(TkRunState v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const TkRunState up{(uint32_t)__shfl_up((int)v.all, o, 64), (uint32_t)__shfl_up((int)v.par, o, 64)};
        if (lane >= o) v = tk_run_combine(up, v);
    }
    return v;
}

// The rounds of ONE piece, by one workgroup (WIDE = false) or by all the workgroups of the launch together (WIDE = true: pieces of
// TK_WIDE_MIN bytes and more -- a megabyte run is a single piece, and one CU walks it at the latency of its 16 wavefronts).  The wide form
// is the same algorithm with "wavefront of the workgroup" read as "wavefront of the grid": the per-wavefront summaries and the round's
// minimum go through global memory, and the workgroup barriers become grid barriers (a counter in global memory; every workgroup of
// the launch is resident: the grid is far smaller than the chip).
#define TK_WIDE_MIN (1u << 17)
#define TK_WIDE_BLOCKS 64
struct TkWideWs {  // workspace of the wide launch (zeroed before it)
    uint32_t bar;      // barrier arrivals (monotonic)
    uint32_t viol;     // a round was not valid: the piece is redone one merge at a time
    uint32_t gmin[2];  // lowest rank of the next round (alternating slots)
    uint32_t sc[4][TK_WIDE_BLOCKS * (TKB_THREADS / 64)];  // per wavefront: run state (all, parity), survivors, odd leading stretch
};
__device__ __forceinline__ uint32_t tk_ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tk_st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all workgroups of the launch: everything written before is visible to everyone after
__device__ __forceinline__ void tk_grid_barrier(uint32_t* bar, uint32_t& epoch) {
    __threadfence();
    __syncthreads();
    epoch += gridDim.x;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();
}

template <bool WIDE>
__device__ __forceinline__ void tk_rounds_piece(const TkTables& T, const uint8_t* __restrict__ text, const uint32_t* ent, uint32_t* g_p0, uint32_t* g_r0,
                                                uint32_t* g_p1, uint32_t* g_r1, TkMiss data, uint32_t* __restrict__ staging,
                                                TkWideWs* ws, uint32_t& epoch, uint32_t* red, uint32_t (*sc_lds)[TKB_THREADS / 64], uint32_t* viol_lds) {
    constexpr int NWB = TKB_THREADS / 64;                               // wavefronts per workgroup
    const uint32_t NWV = WIDE ? (uint32_t)NWB * gridDim.x : (uint32_t)NWB;  // wavefronts that share the piece
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wib = tid >> 6, wid = WIDE ? blockIdx.x * NWB + wib : wib;
    const uint32_t gtid = WIDE ? blockIdx.x * TKB_THREADS + tid : tid, gthreads = WIDE ? gridDim.x * TKB_THREADS : (uint32_t)TKB_THREADS;
    constexpr uint32_t MARK = 0x80000000u;  // (token ids stay below 2^31: checked at tk_create)
    auto sync = [&]() {
        if constexpr (WIDE) tk_grid_barrier(&ws->bar, epoch);
        else __syncthreads();
    };
    auto sc_put = [&](int k, uint32_t v) {
        if constexpr (WIDE) tk_st_agent(&ws->sc[k][wid], v);
        else sc_lds[k][wid] = v;
    };
    auto sc_get = [&](int k, uint32_t q) -> uint32_t {
        if constexpr (WIDE) return tk_ld_agent(&ws->sc[k][q]);
        else return sc_lds[k][q];
    };
    auto set_viol = [&]() {
        if constexpr (WIDE) tk_st_agent(&ws->viol, 1u);
        else *viol_lds = 1;
    };
    // minimum over everyone who shares the piece (round = which of the two global slots; ends with a barrier)
    auto all_min = [&](uint32_t v, uint32_t round) -> uint32_t {
        v = tk_wave_min_u32(v);
        if (lane == 0) red[wib] = v;
        __syncthreads();
        v = tk_wave_min_u32(red[lane & (NWB - 1)]);
        if constexpr (WIDE) {
            if (tid == 0) __hip_atomic_fetch_min(&ws->gmin[round & 1u], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tk_grid_barrier(&ws->bar, epoch);
            v = tk_ld_agent(&ws->gmin[round & 1u]);
        } else
            __syncthreads();
        return v;
    };
    const uint32_t mi = ent[0], s = ent[1], n = ent[2];
    uint32_t *P0 = g_p0 + ent[3], *R0 = g_r0 + ent[3], *P1 = g_p1 + ent[3], *R1 = g_r1 + ent[3];
    if constexpr (WIDE) {  // (the slots were left by the previous piece; nobody uses them before the barrier below)
        if (gtid == 0) {
            tk_st_agent(&ws->viol, 0u);
            tk_st_agent(&ws->gmin[0], TK_RANK_MAX);
            tk_st_agent(&ws->gmin[1], TK_RANK_MAX);
        }
        tk_grid_barrier(&ws->bar, epoch);
    } else {
        if (tid == 0) *viol_lds = 0;
    }
    uint32_t m = TK_RANK_MAX;  // the lowest rank among the pairs: found while the ranks are written (no pass of its own)
    for (uint32_t k = gtid; k < n; k += gthreads) {
        const uint32_t b0 = text[s + k];
        P0[k] = T.byte_rank[b0];
        const uint32_t r = k + 1 < n ? T.pair2[(b0 << 8) | text[s + k + 1]] : TK_RANK_MAX;
        R0[k] = r;
        m = r < m ? r : m;
    }
    uint32_t round = 0;
    m = all_min(m, round);
    uint32_t cnt = n;
    bool redo = false;
    for (;;) {
        if (m == TK_RANK_MAX) break;
        // Every wavefront owns a contiguous range of the parts and walks it in rows of 512 (eight consecutive parts per lane: coalesced,
        // and one wave scan per row); the state that crosses lanes, rows and wavefronts is the parity of the run of rank-m pairs that
        // ends right before a part.
        constexpr int EPL = 8;
        constexpr uint32_t ROW = 64 * EPL;
        const uint32_t per = ((cnt + NWV - 1) / NWV + ROW - 1u) / ROW * ROW;
        const uint32_t wlo = wid * per < cnt ? wid * per : cnt, whi = wlo + per < cnt ? wlo + per : cnt;
        // the parts of a lane: their ranks, rank-m flags (parts beyond the range: none) and the lane's run state
        auto lane_state = [&](uint32_t i0, uint32_t rk[EPL], uint32_t& fm) -> TkRunState {
            if (i0 + EPL <= whi) {  // (ranges and scratch offsets are multiples of four parts: 16-byte loads)
#pragma unroll
                for (int q = 0; q < EPL; q += 4) {
                    const uint4 x = *(const uint4*)(R0 + i0 + q);
                    rk[q] = x.x; rk[q + 1] = x.y; rk[q + 2] = x.z; rk[q + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < EPL; ++q) rk[q] = i0 + q < whi ? R0[i0 + q] : 0u;
            }
            TkRunState v{1u, 0u};
            fm = 0;
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                const bool in = i0 + q < whi;
                const uint32_t f = in ? (uint32_t)(rk[q] == m) : 0u;
                fm |= f << q;
                if (in) v = tk_run_combine(v, TkRunState{f, f});
            }
            return v;
        };
        // b. one pass gives, per wavefront range: its run state, the number of parts that survive (a part right after an odd count of
        // rank-m pairs is absorbed) if no run enters the range, and the length of the run of rank-m pairs the range starts with -- a run
        // of odd length that enters flips the fate of exactly those parts and the one after them
        uint32_t keep0 = 0, lead = 0;
        TkRunState acc{1u, 0u};
        {
            bool open = true;
            for (uint32_t r0 = wlo; r0 < whi; r0 += ROW) {
                const uint32_t i0 = r0 + (uint32_t)EPL * lane;
                uint32_t rk[EPL], fm;
                TkRunState v = tk_run_scan_wave(lane_state(i0, rk, fm), lane);
                TkRunState ex{(uint32_t)__shfl_up((int)v.all, 1, 64), (uint32_t)__shfl_up((int)v.par, 1, 64)};
                if (lane == 0) ex = TkRunState{1u, 0u};
                uint32_t c = tk_run_combine(acc, ex).par, kl = 0;  // parity of the run of rank-m pairs right before the lane's first part
#pragma unroll
                for (int q = 0; q < EPL; ++q) {
                    kl += (uint32_t)(i0 + q < whi) & (c ^ 1u);
                    c = ((fm >> q) & 1u) ? (c ^ 1u) : 0u;
                }
                keep0 += tk_wave_sum_u32(kl);
                if (open) {  // (parts beyond the range have no flag: the leading run ends there at the latest)
                    const uint64_t stop = __ballot(fm != (1u << EPL) - 1u);
                    if (stop) {
                        const int l0 = __ffsll((unsigned long long)stop) - 1;
                        const uint32_t f0 = (uint32_t)__shfl((int)fm, l0, 64);
                        lead += (uint32_t)l0 * EPL + (uint32_t)__ffs((int)~f0) - 1u;
                        open = false;
                    } else
                        lead += ROW;
                }
                acc = tk_run_combine(acc, TkRunState{(uint32_t)__shfl((int)v.all, 63, 64), (uint32_t)__shfl((int)v.par, 63, 64)});
            }
        }
        if (lane == 0) {
            const uint32_t len = whi - wlo, aff = lead + 1u < len ? lead + 1u : len;
            sc_put(0, acc.all);
            sc_put(1, acc.par);
            sc_put(2, keep0);
            sc_put(3, aff & 1u);
        }
        if constexpr (WIDE) {
            if (gtid == 0) tk_st_agent(&ws->gmin[(round + 1u) & 1u], TK_RANK_MAX);  // (last read a round ago, next written after the barrier below)
        }
        sync();
        TkRunState before{1u, 0u};
        uint32_t at = 0, total = 0;
        if constexpr (WIDE) {
            // prefix over the wavefronts of the grid: 64 lanes take a block of summaries each, then one wave scan
            const uint32_t blk = (NWV + 63u) / 64u;
            TkRunState mine{1u, 0u};
            uint32_t k0 = 0, k1 = 0;  // survivors of my summaries if the state that enters my block has parity 0 / 1 (they differ only
                                      // while the block so far is all rank-m pairs)
            for (uint32_t j = 0; j < blk; ++j) {
                const uint32_t q = (uint32_t)lane * blk + j;
                if (q < NWV) {
                    const uint32_t a = sc_get(0, q), pr = sc_get(1, q), kp = sc_get(2, q), od = sc_get(3, q);
                    k0 += kp - (mine.par & od);
                    k1 += kp - ((mine.all ? (mine.par ^ 1u) : mine.par) & od);
                    mine = tk_run_combine(mine, TkRunState{a, pr});
                }
            }
            // exclusive scan of the lanes' states
            TkRunState inc = tk_run_scan_wave(mine, lane);
            TkRunState ex{(uint32_t)__shfl_up((int)inc.all, 1, 64), (uint32_t)__shfl_up((int)inc.par, 1, 64)};
            if (lane == 0) ex = TkRunState{1u, 0u};
            const uint32_t kl = ex.par ? k1 : k0;  // survivors of my block given what enters it
            const uint32_t kinc = tk_wave_scan_u32(kl, lane);
            total = (uint32_t)__shfl((int)kinc, 63, 64);
            // my own wavefront's place: the lane that holds summary `wid`, then the summaries of its block before it
            const int owner = (int)(wid / blk);
            TkRunState st{(uint32_t)__shfl((int)ex.all, owner, 64), (uint32_t)__shfl((int)ex.par, owner, 64)};
            at = (uint32_t)__shfl((int)(kinc - kl), owner, 64);
            for (uint32_t q = (uint32_t)owner * blk; q < wid; ++q) {
                at += sc_get(2, q) - (st.par & sc_get(3, q));
                st = tk_run_combine(st, TkRunState{sc_get(0, q), sc_get(1, q)});
            }
            before = st;
        } else {
            TkRunState st{1u, 0u};
            for (uint32_t q = 0; q < NWV; ++q) {
                if (q == wid) before = st;
                const uint32_t k = sc_get(2, q) - (st.par & sc_get(3, q));
                if (q < wid) at += k;
                total += k;
                st = tk_run_combine(st, TkRunState{sc_get(0, q), sc_get(1, q)});
            }
        }
        // c. the survivors move to their places
        {
            TkRunState carry = before;
            for (uint32_t r0 = wlo; r0 < whi; r0 += ROW) {
                const uint32_t i0 = r0 + (uint32_t)EPL * lane;
                uint32_t rk[EPL], fm;
                TkRunState v = tk_run_scan_wave(lane_state(i0, rk, fm), lane);
                TkRunState ex{(uint32_t)__shfl_up((int)v.all, 1, 64), (uint32_t)__shfl_up((int)v.par, 1, 64)};
                if (lane == 0) ex = TkRunState{1u, 0u};
                uint32_t c = tk_run_combine(carry, ex).par, kl = 0, keepm = 0;
#pragma unroll
                for (int q = 0; q < EPL; ++q) {
                    const uint32_t k = (uint32_t)(i0 + q < whi) & (c ^ 1u);
                    keepm |= k << q;
                    kl += k;
                    c = ((fm >> q) & 1u) ? (c ^ 1u) : 0u;
                }
                const uint32_t inc = tk_wave_scan_u32(kl, lane);
                uint32_t o = at + inc - kl;
#pragma unroll
                for (int q = 0; q < EPL; ++q) {
                    if ((keepm >> q) & 1u) {  // kept; selected when its own pair has rank m
                        const uint32_t i = i0 + q, f = (fm >> q) & 1u;
                        P1[o] = f ? (m | MARK) : (P0[i] & ~MARK);
                        R1[o] = rk[q];  // (still right when neither this part nor the next one changes)
                        ++o;
                        // the next pick of this round is the pair right after this one: between the two picks the pair (merged part,
                        // its still unmerged right neighbour) exists and must not rank below m either
                        if (f && i + 2 < cnt && R0[i + 2] == m && tk_probe_pair(T, m, P0[i + 2] & ~MARK) < m) set_viol();
                    }
                }
                at += (uint32_t)__shfl((int)inc, 63, 64);
                carry = tk_run_combine(carry, TkRunState{(uint32_t)__shfl((int)v.all, 63, 64), (uint32_t)__shfl((int)v.par, 63, 64)});
            }
        }
        sync();
        // d. ranks of the pairs that a merge has touched; the lowest rank of the next round on the way
        uint32_t mn = TK_RANK_MAX;
        for (uint32_t j = gtid; j < total; j += gthreads) {
            const uint32_t a = P1[j], b = j + 1 < total ? P1[j + 1] : 0u;
            uint32_t r;
            if (j + 1 >= total) R1[j] = r = TK_RANK_MAX;
            else if ((a | b) & MARK) {
                r = tk_probe_pair(T, a & ~MARK, b & ~MARK);
                R1[j] = r;
                if (r < m) set_viol();  // a merge created a pair that the reference would have picked before the rest of this round
            } else
                r = R1[j];
            mn = r < mn ? r : mn;
        }
        ++round;
        m = all_min(mn, round);  // (its barrier also publishes the violation flag and this round's writes)
        if (WIDE ? tk_ld_agent(&ws->viol) : *viol_lds) {
            redo = true;
            break;
        }
        uint32_t* t0 = P0; P0 = P1; P1 = t0;
        t0 = R0; R0 = R1; R1 = t0;
        cnt = total;
    }
    if (redo) {
        if (gtid == 0) tk_put_result(data, mi, TK_MERGE_REDO, 0u);
    } else {
        for (uint32_t k = gtid; k < cnt; k += gthreads) staging[s + k] = P0[k] & ~MARK;
        if (gtid == 0) tk_put_result(data, mi, cnt, cnt == 1 ? (P0[0] & ~MARK) : s);
    }
    sync();
}

// one workgroup per piece (pieces of TK_WIDE_MIN bytes and more are left to tk_k_merge_rounds_wide)
__global__ __launch_bounds__(TKB_THREADS) void tk_k_merge_rounds(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ listC,
                                                                  uint32_t nC, uint32_t* __restrict__ g_p0, uint32_t* __restrict__ g_r0,
                                                                  uint32_t* __restrict__ g_p1, uint32_t* __restrict__ g_r1,
                                                                  TkMiss data, uint32_t* __restrict__ staging) {
    __shared__ uint32_t red[TKB_THREADS / 64];
    __shared__ uint32_t sc[4][TKB_THREADS / 64];
    __shared__ uint32_t viol_sh;
    uint32_t epoch = 0;
    for (uint32_t w = blockIdx.x; w < nC; w += gridDim.x) {
        const uint32_t* ent = listC + 5 * (uint64_t)w;
        if (ent[2] >= TK_WIDE_MIN) continue;
        tk_rounds_piece<false>(T, text, ent, g_p0, g_r0, g_p1, g_r1, data, staging, nullptr, epoch, red, sc, &viol_sh);
    }
}
// all workgroups of the launch (TK_WIDE_BLOCKS of them) on one piece after the other
__global__ __launch_bounds__(TKB_THREADS) void tk_k_merge_rounds_wide(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ listC,
                                                                       uint32_t nC, uint32_t* __restrict__ g_p0, uint32_t* __restrict__ g_r0,
                                                                       uint32_t* __restrict__ g_p1, uint32_t* __restrict__ g_r1,
                                                                       TkMiss data, uint32_t* __restrict__ staging, TkWideWs* __restrict__ ws) {
    __shared__ uint32_t red[TKB_THREADS / 64];
    uint32_t epoch = 0;
    for (uint32_t w = 0; w < nC; ++w) {
        const uint32_t* ent = listC + 5 * (uint64_t)w;
        if (ent[2] < TK_WIDE_MIN) continue;
        tk_rounds_piece<true>(T, text, ent, g_p0, g_r0, g_p1, g_r1, data, staging, ws, epoch, red, nullptr, nullptr);
    }
}

// ------------------------------------------------------------------------------------------
// back end: tk_k_count_tiles -> exclusive scan of the tile counts -> tk_k_place (and, beside it, tk_k_docoff)
//
// The tokens of tile t go behind those of all the tiles before it, and a tile's token count is known only when the results of its
// missed pieces are (a missed piece is two or more tokens).  Both passes read the per-piece result words -- four pieces per lane and
// row of 256, one 16-byte load -- and, for a missed piece, the entry its word refers to (TkMiss): the first takes the count from it,
// the second the tokens themselves, which a table entry holds inline (one random access per occurrence; earlier rounds copied every
// result into a per-tile array first and moved 5 GB per GiB of text to place 1 GB of tokens).
// Measured and dropped in round 4 (profiles/r04_place_single_pass.txt): ONE pass with a decoupled look-back over the tile counts --
// 2.3 ms where these two passes take less, because a wavefront holds its tile while it waits for the tiles before it, and a kernel
// whose workgroups wait for each other cannot share the device with another kernel (a chunk's back stage runs beside the next chunk's
// front stage): its workgroups that are not resident yet never get the registers the waiting ones hold.
// ------------------------------------------------------------------------------------------

// rows of 256 pieces a wavefront has in flight together (a tile of web text has 580 pieces): three for chunks, one for small inputs
// (template parameter ROWS: a kernel that runs once over a few tiles pays for the code it has to fetch, and the three-row form is three
// times the code -- 96 us for a 4 KiB call)
#ifndef TKP_ROWS_COUNT
#define TKP_ROWS_COUNT 3
#endif
#ifndef TKP_ROWS_PLACE
#define TKP_ROWS_PLACE 1
#endif

// tile_nt[t] <- tokens of tile t = its pieces - its gap chars + what its pieces that are not tokens have beyond one token each; total[1] +=
// pieces of the chunk.  One wavefront per tile at a time.  Round 4, second form: the front kernel leaves the entries of a tile's missed pieces
// once more at the tail of the tile's run of result words (TKF_TAIL_*), so this pass reads 0.12 GB per GiB instead of every result word
// (0.65 GB), four refs per lane and their count bytes (an L2-resident array) in flight together.  (The sums of a tile's rows of 256 pieces,
// which the document offsets need, come from tk_k_place now.)
template <int ROWS>
__global__ __launch_bounds__(256) void tk_k_count_tiles(uint64_t ntiles, const uint32_t* __restrict__ tile_np, const uint32_t* __restrict__ res, TkMiss data,
                                                        uint32_t* __restrict__ tile_nt, unsigned long long* __restrict__ total) {
    // HALF a wavefront per tile (a tile of web text has ~110 missed pieces; the pass is a chain of four dependent loads per tile, so two tiles
    // per wavefront are twice the tiles in flight); every half runs the same number of rounds (the sums are wave-wide instructions)
    const int lane = threadIdx.x & 63;
    const uint32_t hl = (uint32_t)lane & 31u;
    const uint64_t half = (blockIdx.x * 256ull + threadIdx.x) >> 5, nhalves = ((uint64_t)gridDim.x * 256) >> 5;
    const uint64_t rounds = (ntiles + nhalves - 1) / nhalves;
    unsigned long long pieces = 0;
    uint32_t np_next = half < ntiles ? tile_np[half] : 0u;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t t = half + r * nhalves;
        const bool have = t < ntiles;
        const uint32_t np = np_next, rb = have ? (uint32_t)t * TKF_CAP : 0u;
        np_next = t + nhalves < ntiles ? tile_np[t + nhalves] : 0u;  // (the next tile's size is on its way while this tile is counted)
        if (hl == 0) pieces += np;
        uint2 tail = make_uint2(0u, 0u);  // {gap chars, missed pieces}
        if (np) tail = *(const uint2*)(res + rb + TKF_TAIL_NGAP);
        const uint32_t ng = tail.x, nm = tail.y;
        uint32_t extra = 0;
        const uint32_t nm_max = max(nm, (uint32_t)__shfl_xor((int)nm, 32, 64));  // (both halves walk the longer list)
        for (uint32_t q0 = 0; q0 < nm_max; q0 += 32u * ROWS * 2u) {
            uint32_t ref[ROWS * 2], cb[ROWS * 2];
#pragma unroll
            for (int j = 0; j < ROWS * 2; ++j) {
                const uint32_t q = q0 + (uint32_t)j * 32u + hl;
                ref[j] = res[rb + TKF_TAIL_REFS - (q < nm ? q : 0u)];
                if (q >= nm) ref[j] = 0xFFFFFFFFu;  // (also what the front kernel leaves for a piece it had no entry for: one token)
            }
            bool escape = false;
#pragma unroll
            for (int j = 0; j < ROWS * 2; ++j) {
                const bool in_tab = ref[j] < data.ovf_base;
                cb[j] = data.cnt8[in_tab ? ref[j] : 0u];
                if (ref[j] != 0xFFFFFFFFu && (!in_tab || cb[j] == 255u)) escape = true;
                else if (in_tab) extra += cb[j] - 1u;
            }
            if (__ballot(escape)) {  // (an overflow entry, a piece of 255 tokens and more: the entry itself)
#pragma unroll
                for (int j = 0; j < ROWS * 2; ++j) {
                    const bool in_tab = ref[j] < data.ovf_base, esc = ref[j] != 0xFFFFFFFFu && (!in_tab || cb[j] == 255u);
                    const uint32_t cw = data.head(esc ? ref[j] : data.ovf_base)[2];
                    if (esc) extra += TKD_COUNT(cw) - 1u;
                }
            }
        }
        extra = tk_row16_sum(extra);                                   // the sixteen lanes of a DPP row
        extra += (uint32_t)__shfl_xor((int)extra, 16, 64);             // ... and the other row of the half
        if (have && hl == 0) tile_nt[t] = np - ng + extra;
    }
    // (one atomic per WORKGROUP: same-address atomics are served one after the other at ~10 M/s on this part -- one per wavefront of a
    // 4096-workgroup grid was 0.2 ms, the floor of this kernel however small the chunk)
    __shared__ unsigned long long pieces_sh[8];
    if (hl == 0) pieces_sh[threadIdx.x >> 5] = pieces;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long p = 0;
        for (int i = 0; i < 8; ++i) p += pieces_sh[i];
        if (p) atomicAdd(&total[1], p);
    }
}

// tile_tb = exclusive scan of tile_nt.  One wavefront per tile at a time, 256 * ROWS pieces per step.  Round 5, rebuilt twice over:
//  * the pieces that are not single tokens -- a fifth of them on web text, and gap chars -- are taken out of the rows into a LIST (in
//    piece order) and handled by dense lanes: head {count, first token} and the entry's next sixteen bytes in one go, a prefix sum of
//    `count - 1` over the list, the tokens to their place.  A piece's place is its index plus the sum of `count - 1` over the listed pieces
//    before it, so the single tokens of the rows need nothing but that sum (an LDS word the dense lanes leave).  Round 4 kept four pieces
//    per lane and row side by side: twelve head loads, twelve entry loads and twelve times thirteen predicated stores per lane and step,
//    115 registers;
//  * a tile's tokens are put together in LDS and leave in whole lines when there are at most TKP_STAGE of them (a tile of web text has
//    ~900; other tiles write them lane by lane as before): a lane's own stores are 4 bytes every 16, each wavefront instruction a
//    partial write of sixteen lines -- 136 M write requests per GiB where the tokens need 16 M (TCP_TCC_WRITE_REQ, profiles/r04_sq_counters.csv).
// 1.21 -> 0.78 ms per GiB.  Neither half alone moved it (the list without the staging: 1.14 ms; round 4's kernel with every store sent to one
// 4 KiB window: 1.06), nor did more rows in flight or a staging per step, which loads every head twice (0.97): profiles/r05_place_experiments.txt.
#ifndef TKP_STAGE
#define TKP_STAGE 1024
#endif
struct TkPlaceDocs {
    const uint64_t* doc_off;    // [n_docs + 1] offsets of the chunk's documents (absolute: chunk_base is subtracted)
    uint64_t chunk_base, n, n_docs;
    const uint32_t* doc_first;  // [ntiles + 1] tk_k_mark_docs: first document that starts in the tile (0xFFFFFFFF: none); [ntiles]: at or behind the end
    const uint32_t* starts;     // piece-start bitmap, 120 words per tile
    const uint64_t* total;      // [0]: the chunk's token count
    uint64_t* tok_off;          // [n_docs + 1] out (null: not asked for)
};
template <int ROWS>
__global__ __launch_bounds__(256, 6) void tk_k_place(uint64_t ntiles, const uint32_t* __restrict__ tile_np, const uint32_t* __restrict__ tile_tb,
                                                  const uint32_t* __restrict__ res, TkMiss data, const uint32_t* __restrict__ staging, uint32_t* __restrict__ out_all,
                                                  const unsigned long long* __restrict__ tok_base, uint32_t* __restrict__ big,
                                                  // (round 6) the document offsets, written here as a tile's pieces are placed: tok_off[d] = tokens before the piece
                                                  // that starts document d.  tk_k_docoff found that piece again per document -- five dependent loads, the result
                                                  // words of up to 255 pieces before it read once more: 0.15 ms and 0.36 GB per GiB.  docs.tok_off == nullptr: not asked for.
                                                  TkPlaceDocs docs) {
    constexpr uint32_t CAP = 256u * ROWS, SCAP = (uint32_t)TKP_STAGE;
    __shared__ __attribute__((aligned(16))) uint32_t stage_sh[4][SCAP + 4];
    __shared__ uint32_t ref_sh[4][CAP];  // the listed pieces' result words, in piece order
    __shared__ uint32_t cum_sh[4][CAP];  // sum of `count - 1` over the list up to and including each of them
    __shared__ uint16_t idx_sh[4][CAP];  // their index among the step's pieces
    // the chunk's tokens follow those of the chunks before it: their number stays on the device (chunks are pipelined, the host does
    // not know it when it queues this kernel)
    uint32_t* __restrict__ out = out_all + tok_base[0];
    const int lane = threadIdx.x & 63;
    uint32_t* refl = ref_sh[threadIdx.x >> 6];
    uint32_t* cuml = cum_sh[threadIdx.x >> 6];
    uint16_t* idxl = idx_sh[threadIdx.x >> 6];
    uint32_t* stl = stage_sh[threadIdx.x >> 6];
    // (the wavefront's index as a scalar: the tile's index, its sizes and bases and everything derived from them live in scalar registers -- with
    // the document offsets written here the kernel had grown from 74 to 98 vector registers, from six workgroups per CU to four)
    const uint64_t wave = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6)), nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    uint32_t np_next = wave < ntiles ? tile_np[wave] : 0u, tb_next = wave < ntiles ? tile_tb[wave] : 0u;
    uint32_t te_next = wave + 1 < ntiles ? tile_tb[wave + 1] : 0xFFFFFFFFu;  // where the tile's tokens end (the chunk's last tile: unknown here, never staged)
    const bool want_docs = docs.tok_off != nullptr;
    uint32_t df_next = (want_docs && wave < ntiles) ? docs.doc_first[wave] : 0xFFFFFFFFu;
    const uint64_t tok_base_global = tok_base[0];
    for (uint64_t t = wave; t < ntiles; t += nwaves) {
        const uint32_t np = np_next, rb = (uint32_t)t * TKF_CAP, run0 = tb_next;  // (token offsets within a chunk fit 32 bits: a chunk is less than 4 GiB of text)
        const uint32_t nt_tile = te_next - run0;
        const uint32_t dfirst = df_next;
        np_next = t + nwaves < ntiles ? tile_np[t + nwaves] : 0u;  // (the next tile's size and base are on their way while this tile is placed)
        tb_next = t + nwaves < ntiles ? tile_tb[t + nwaves] : 0u;
        te_next = t + nwaves + 1 < ntiles ? tile_tb[t + nwaves + 1] : 0xFFFFFFFFu;
        df_next = (want_docs && t + nwaves < ntiles) ? docs.doc_first[t + nwaves] : 0xFFFFFFFFu;
        // The documents that start in this tile: the piece index kp of each (its start is a piece start: the number of set bits of the tile's
        // start bitmap before it), for the first sixty-four of them before the tile's steps (a tile of web text has one or two), further ones
        // -- a tile full of tiny documents -- inside every step.  A document's offset is written by the step that holds its piece.
        const bool has_docs = dfirst != 0xFFFFFFFFu;  // (wave-uniform)
        const uint64_t tile_lo = t * (uint64_t)TK_TILE, tile_hi = tile_lo + TK_TILE < docs.n ? tile_lo + TK_TILE : docs.n;
        // kp of document dfirst + 64 c + lane, or 0xFFFFFFFF when it does not start in this tile (every lane takes part: the shuffles read all lanes).
        // The lane's two words of the start bitmap and the set bits before them are loaded where they are needed (once per tile with documents; a
        // tile with more than sixty-four loads them again in every step) and are not carried through the tile's steps.
        auto doc_piece = [&](uint32_t c) -> uint32_t {
            uint32_t sb0 = 0, sb1 = 0;
            if (lane < (int)(TK_TILE / 64)) {
                const uint2 sw = *(const uint2*)(docs.starts + t * (TK_TILE / 32) + 2u * (uint32_t)lane);
                sb0 = sw.x;
                sb1 = sw.y;
            }
            const uint32_t cb = (uint32_t)__popc(sb0) + (uint32_t)__popc(sb1);
            const uint32_t spx = tk_wave_scan_u32(cb, lane) - cb;
            const uint64_t d = (uint64_t)dfirst + 64u * c + (uint32_t)lane;
            const uint64_t pos = d < docs.n_docs ? docs.doc_off[d] - docs.chunk_base : ~0ull;
            const bool in = pos >= tile_lo && pos < tile_hi;
            const uint32_t it = in ? (uint32_t)(pos - tile_lo) : 0u, w = it >> 5;
            const uint32_t a0 = (uint32_t)__shfl((int)sb0, (int)(w >> 1), 64), a1 = (uint32_t)__shfl((int)sb1, (int)(w >> 1), 64);
            const uint32_t px = (uint32_t)__shfl((int)spx, (int)(w >> 1), 64);
            const uint32_t kp = px + ((w & 1u) ? (uint32_t)__popc(a0) + (uint32_t)__popc(a1 & ((1u << (it & 31u)) - 1u)) : (uint32_t)__popc(a0 & ((1u << (it & 31u)) - 1u)));
            return in ? kp : 0xFFFFFFFFu;
        };
        const uint32_t kp0 = has_docs ? doc_piece(0u) : 0xFFFFFFFFu;
        const bool more_docs = has_docs && __ballot(kp0 != 0xFFFFFFFFu) == ~0ull;  // (all sixty-four start here: there may be more)
        auto place_tile = [&](auto staged_c) {
            constexpr bool ST = decltype(staged_c)::value;
            auto W = [&](uint32_t off, uint32_t v) {  // token number `off` of the tile
                if constexpr (ST) stl[off] = v;
                else out[run0 + off] = v;
            };
            uint32_t run = 0;  // tokens of the tile's steps so far
            for (uint32_t k0 = 0; k0 < np; k0 += CAP) {
                // the step's result words: four pieces per lane and row (no `if` around the loads: the rows are in flight together)
                uint32_t tk[ROWS][4];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const uint32_t k = k0 + (uint32_t)r * 256u + (uint32_t)lane * 4u;
                    const uint4 t4 = *(const uint4*)(res + rb + (k < np ? k : 0u));
                    tk[r][0] = k < np ? t4.x : TK_RES_GAP;  // (dead words of a run's last four count as "no token")
                    tk[r][1] = k + 1 < np ? t4.y : TK_RES_GAP;
                    tk[r][2] = k + 2 < np ? t4.z : TK_RES_GAP;
                    tk[r][3] = k + 3 < np ? t4.w : TK_RES_GAP;
                }
                // the list: pieces that are not one token (a reference to an entry; no token at all), row by row = in piece order
                uint32_t fm[ROWS], lbase[ROWS], nlist = 0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    fm[r] = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((tk[r][j] & TK_RES_FLAG) || tk[r][j] == TK_RES_GAP) fm[r] |= 1u << j;
                    const uint32_t nf = (uint32_t)__popc(fm[r]);
                    const uint32_t inc = tk_wave_scan_u32(nf, lane);
                    lbase[r] = nlist + inc - nf;
                    nlist += (uint32_t)__shfl((int)inc, 63, 64);
                    uint32_t at = lbase[r];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((fm[r] >> j) & 1u) {
                            refl[at] = tk[r][j];
                            idxl[at] = (uint16_t)((uint32_t)r * 256u + (uint32_t)lane * 4u + (uint32_t)j);
                            ++at;
                        }
                }
                __builtin_amdgcn_wave_barrier();
                // the listed pieces, sixty-four at a time: entry -> count and tokens, prefix sum of count - 1 (a gap char: minus one), tokens to their place
                uint32_t carry = 0;  // the sum so far (wave-uniform)
                for (uint32_t d0 = 0; d0 < nlist; d0 += 64u) {
                    const uint32_t d = d0 + (uint32_t)lane;
                    const bool have = d < nlist;
                    const uint32_t w = have ? refl[d] : TK_RES_GAP;
                    const uint32_t pidx = have ? (uint32_t)idxl[d] : 0u;
                    const bool flagged = (w & TK_RES_FLAG) != 0u;
                    const uint32_t e = w & ~TK_RES_FLAG;
                    const bool in_tab = flagged && e < data.ovf_base;
                    // (no `if` around the loads: the head and the entry's next sixteen bytes are in flight together; a lane without an entry
                    // reads the first overflow entry)
                    const uint2 hd = data.result(flagged ? e : data.ovf_base);
                    const uint4 a = *(const uint4*)(in_tab ? (const uint32_t*)(data.tab[e].tok + 1) : (const uint32_t*)data.ovf);
                    const uint32_t cnt = flagged ? TKD_COUNT(hd.x) : 0u;
                    const uint32_t extra = have ? cnt - 1u : 0u;
                    const uint32_t incx = tk_wave_scan_u32(extra, lane);
                    const uint32_t cum = carry + incx;
                    if (have) cuml[d] = cum;
                    carry += (uint32_t)__shfl((int)incx, 63, 64);
                    const uint32_t off = run + pidx + (cum - extra);
                    bool stg = false;  // its tokens are in the staging area
                    if (flagged) {
                        if (cnt == 1u) {  // (a long piece that is a token after all, tk_k_bincount; entries written that way)
                            W(off, hd.y);
                        } else if (hd.x & TKD_INLINE_BIT) {
                            W(off, hd.y);
                            W(off + 1u, a.x);
                            if (cnt > 2u) W(off + 2u, a.y);
                            if (cnt > 3u) W(off + 3u, a.z);
                            if (cnt > 4u) W(off + 4u, a.w);
                            if (cnt > 5u) {
                                const uint32_t* tokp = data.tab[e].tok;
                                const uint4 b = *(const uint4*)(tokp + 5);
                                uint4 c4 = make_uint4(0, 0, 0, 0);
                                if (cnt > 9u) c4 = *(const uint4*)(tokp + 9);
                                W(off + 5u, b.x);
                                if (cnt > 6u) W(off + 6u, b.y);
                                if (cnt > 7u) W(off + 7u, b.z);
                                if (cnt > 8u) W(off + 8u, b.w);
                                if (cnt > 9u) {
                                    W(off + 9u, c4.x);
                                    if (cnt > 10u) W(off + 10u, c4.y);
                                    if (cnt > 11u) W(off + 11u, c4.z);
                                    if (cnt > 12u) W(off + 12u, c4.w);
                                }
                            }
                        } else if (cnt > 1u) {
                            stg = true;
                        }
                    }
                    // what does not fit an entry -- more than TKD_INLINE tokens, overflow entries -- is copied from the staging area by the
                    // whole wavefront, a piece at a time (a few per tile); thousands of tokens of one piece by tk_k_bigcopy with the whole device
                    for (uint64_t m = __ballot(stg); m; m &= m - 1ull) {
                        const int l = __ffsll((unsigned long long)m) - 1;
                        const uint32_t src = (uint32_t)__shfl((int)hd.y, l, 64), cc = (uint32_t)__shfl((int)cnt, l, 64), dst = (uint32_t)__shfl((int)off, l, 64);
                        if (ST || cc < TK_BIGCOPY) {  // (a tile that goes through LDS holds no piece that long)
                            for (uint32_t i = (uint32_t)lane; i < cc; i += 64u) W(dst + i, staging[src + i]);
                        } else if (lane == 0) {
                            const uint32_t bat = atomicAdd(&big[0], 1u);
                            if (bat < TK_BIGCOPY_CAP) {
                                big[1 + 3 * bat] = src;
                                big[2 + 3 * bat] = run0 + dst;
                                big[3 + 3 * bat] = cc;
                            } else {
                                for (uint32_t i = 0; i < cc; ++i) out[run0 + dst + i] = staging[src + i];
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (has_docs) {
                    // the documents whose first piece lies in this step: tokens before it = tokens of the steps so far + its index in the step + the
                    // sum of `count - 1` over the listed pieces before it (their number: the list base of the lane that holds it + its flagged
                    // pieces before -- shuffled from that lane)
                    auto emit = [&](uint32_t kp, uint32_t c) {
                        const bool mine = kp != 0xFFFFFFFFu && kp >= k0 && kp < k0 + CAP;
                        const uint32_t j = mine ? kp - k0 : 0u, src = (j & 255u) >> 2;
                        uint32_t lb = 0, fmv = 0;
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) {
                            const uint32_t lbr = (uint32_t)__shfl((int)lbase[r], (int)src, 64), fmr = (uint32_t)__shfl((int)fm[r], (int)src, 64);
                            if ((j >> 8) == (uint32_t)r) {
                                lb = lbr;
                                fmv = fmr;
                            }
                        }
                        const uint32_t nl = lb + (uint32_t)__popc(fmv & ((1u << (j & 3u)) - 1u));
                        const uint32_t ex = (mine && nl) ? cuml[nl - 1u] : 0u;
                        if (mine) docs.tok_off[(uint64_t)dfirst + 64u * c + (uint32_t)lane] = tok_base_global + (uint64_t)(uint32_t)(run0 + run + j + ex);  // (32-bit sum first: `ex` is negative behind gap chars)
                    };
                    emit(kp0, 0u);
                    if (more_docs)
                        for (uint32_t c = 1;; ++c) {
                            const uint32_t kp = doc_piece(c);
                            emit(kp, c);
                            if (__ballot(kp != 0xFFFFFFFFu) != ~0ull) break;
                        }
                }
                // the single tokens of the rows: place = index + the sum over the listed pieces before
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const uint32_t kr = (uint32_t)r * 256u;
                    if (k0 + kr >= np) break;
                    uint32_t rk = lbase[r];
                    uint32_t ex = rk ? cuml[rk - 1u] : 0u;

                    const uint32_t o = run + kr + (uint32_t)lane * 4u;
                    if (fm[r] == 0u) {
                        if constexpr (ST) {
                            W(o + ex, tk[r][0]);
                            W(o + ex + 1u, tk[r][1]);
                            W(o + ex + 2u, tk[r][2]);
                            W(o + ex + 3u, tk[r][3]);
                        } else {
                            // (the sum in 32 bits FIRST: `ex` is negative behind gap chars, as an unsigned number that only works modulo 2^32)
                            *(uint4*)(out + (uint32_t)(run0 + o + ex)) = make_uint4(tk[r][0], tk[r][1], tk[r][2], tk[r][3]);  // (4-byte aligned 16-byte store)
                        }
                    } else if (fm[r] != 15u) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if ((fm[r] >> j) & 1u) ex = cuml[rk++];
                            else W(o + (uint32_t)j + ex, tk[r][j]);
                        }
                    }
                }
                run += CAP + carry;  // (a step's dead words cancel out: one piece, minus one)
                __builtin_amdgcn_wave_barrier();  // (the list is reused by the next step)
            }
            if constexpr (ST) {  // LDS -> the output, sixteen bytes a lane
                for (uint32_t i = (uint32_t)lane * 4u; i < nt_tile; i += 256u) {
                    const uint4 v = *(const uint4*)(stl + i);
                    uint32_t* q = out + run0 + i;
                    if (i + 4u <= nt_tile) {
                        *(uint4*)q = v;  // (4-byte aligned 16-byte store)
                    } else {
                        q[0] = v.x;
                        if (i + 1u < nt_tile) q[1] = v.y;
                        if (i + 2u < nt_tile) q[2] = v.z;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        if (np != 0u && nt_tile <= SCAP) place_tile(std::true_type{});
        else place_tile(std::false_type{});
        if (want_docs && t + 1 == ntiles) {  // documents at or behind the end of the text (empty ones), and the closing offset: the chunk's token count
            const uint32_t dl = docs.doc_first[ntiles];
            const uint64_t v = tok_base_global + docs.total[0];
            for (uint64_t d = (dl == 0xFFFFFFFFu ? docs.n_docs : (uint64_t)dl) + (uint32_t)lane; d <= docs.n_docs; d += 64u) docs.tok_off[d] = v;
        }
    }
}

// the token runs of very long pieces (big[0] entries {staging position, output position, count} recorded by tk_k_place): every entry is
// copied by the whole grid
__global__ __launch_bounds__(256) void tk_k_bigcopy(const uint32_t* __restrict__ big, const uint32_t* __restrict__ staging, uint32_t* __restrict__ out_all,
                                                    const unsigned long long* __restrict__ tok_base) {
    uint32_t* __restrict__ out = out_all + tok_base[0];
    const uint32_t nb = big[0] < TK_BIGCOPY_CAP ? big[0] : (uint32_t)TK_BIGCOPY_CAP;
    for (uint32_t e = 0; e < nb; ++e) {
        const uint32_t src = big[1 + 3 * e], dst = big[2 + 3 * e], cc = big[3 + 3 * e];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < cc; i += gridDim.x * 256u) out[dst + i] = staging[src + i];
    }
}

// tok_off[d] = tokens before the piece at which document d starts.  SIXTEEN LANES per document (round 4; a wavefront per document before:
// the kernel waits for a chain of five dependent loads per document, so four documents per wavefront are four times the documents in flight).
// The piece is found in the piece-start bitmap of the document's tile (a document start is a hard piece start); its token offset is the tile's
// place (tile_tb) plus what tk_k_count_tiles has left for the row of 256 pieces it lies in plus the counts of the row's pieces before it --
// single tokens count one, the others what their count byte says.  (Needs nothing of tk_k_place.)
__global__ __launch_bounds__(256) void tk_k_docoff(uint64_t n_docs, const uint64_t* __restrict__ doc_off, uint64_t chunk_base, uint64_t n,
                                                    const uint32_t* __restrict__ starts, const uint32_t* __restrict__ tile_tb,
                                                    const uint32_t* __restrict__ res, TkMiss data, const uint32_t* __restrict__ row_rel /* tk_k_place's row_abs */,
                                                    const uint64_t* __restrict__ total, const unsigned long long* __restrict__ tok_base, uint64_t* __restrict__ tok_off) {
    const uint64_t tok_base_global = tok_base[0];  // tokens of the chunks before this one
    const uint32_t sl = threadIdx.x & 15u;
    const uint64_t grp = (blockIdx.x * 256ull + threadIdx.x) >> 4, ngrp = ((uint64_t)gridDim.x * 256) >> 4;
    const uint64_t rounds = (n_docs + 1 + ngrp - 1) / ngrp;  // (every group runs the same number of rounds: the row sums are wave-wide instructions)
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t d = grp + r * ngrp;
        const bool have = d <= n_docs;
        const uint64_t pos = (have && d < n_docs) ? doc_off[d] - chunk_base : n;
        const bool inside = pos < n;  // (else: empty documents at the end of the chunk, and the closing offset -> the chunk's total)
        const uint32_t t = (inside && row_rel) ? (uint32_t)(pos / TK_TILE) : 0u, in_tile = inside ? (uint32_t)(pos - (uint64_t)t * TK_TILE) : 0u;  // (no rows: one piece)
        // pieces of the tile that start before pos: eight words of the tile's 120 per lane
        uint32_t kp = 0;
        if (row_rel) {
            const uint32_t* sw = starts + (uint64_t)t * (TK_TILE / 32) + sl * 8u;
            const bool any = sl * 256u < in_tile;
            const uint4 a = *(const uint4*)(any ? sw : starts), b = *(const uint4*)(any && sl * 256u + 128u < in_tile ? sw + 4 : starts);
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t lo = sl * 256u + (uint32_t)j * 32u;
                if (lo < in_tile) kp += (uint32_t)__popc(in_tile - lo >= 32u ? w[j] : (w[j] & ((1u << (in_tile - lo)) - 1u)));
            }
            kp = tk_row16_sum(kp);
        }
        const uint32_t rb = t * TKF_CAP;
        uint64_t row_run = tile_tb[t];
        uint32_t kstart = 0;
        if (row_rel && kp >= 256u) {  // (the row's place in the chunk, left by tk_k_place: the document's first piece is piece kp of the tile, so the row exists)
            row_run = row_rel[(uint64_t)t * (TKF_CAP / 256) + (kp >> 8)];
            kstart = kp & ~255u;
        }
        // the row's pieces before the document's: at most 255, sixteen per lane, all loads without an `if` of their own
        uint32_t sum = 0;
        {
            const uint32_t k0 = kstart + sl * 16u;
            const uint32_t* rp = res + rb + (k0 < kp ? k0 : kstart);
            const uint4 q0 = ((const uint4*)rp)[0], q1 = ((const uint4*)rp)[1], q2 = ((const uint4*)rp)[2], q3 = ((const uint4*)rp)[3];
            uint32_t rv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            uint32_t cb[16];
            bool escape = false;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (k0 + (uint32_t)j >= kp) rv[j] = TK_RES_GAP;
                const bool flagged = rv[j] != TK_RES_GAP && (rv[j] & TK_RES_FLAG), in_tab = flagged && (rv[j] & ~TK_RES_FLAG) < data.ovf_base;
                cb[j] = data.cnt8[in_tab ? (rv[j] & ~TK_RES_FLAG) : 0u];
                if (flagged && (!in_tab || cb[j] == 255u)) escape = true;
                else sum += flagged ? cb[j] : (rv[j] != TK_RES_GAP ? 1u : 0u);
            }
            if (escape) {  // (rare: an overflow entry, a piece of 255 tokens and more)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool flagged = rv[j] != TK_RES_GAP && (rv[j] & TK_RES_FLAG), in_tab = flagged && (rv[j] & ~TK_RES_FLAG) < data.ovf_base;
                    if (flagged && (!in_tab || cb[j] == 255u)) sum += TKD_COUNT(data.head(rv[j] & ~TK_RES_FLAG)[2]);
                }
            }
        }
        const uint64_t v = inside ? row_run + tk_row16_sum(sum) : total[0];
        if (have && sl == 0) tok_off[d] = tok_base_global + v;
    }
}

// chunk k knows its token count: the next chunk's tokens start behind them (tok_bases[0] = 0)
__global__ void tk_k_advance(unsigned long long* __restrict__ tok_bases, uint32_t k, const uint64_t* __restrict__ total) {
    if (blockIdx.x == 0 && threadIdx.x == 0) tok_bases[k + 1] = tok_bases[k] + total[0];
}

// encode_single_piece (src/py.rs:145-150): the whole buffer is one piece, no pre-tokenisation
// (no_lookup: byte_pair_encode proper, src/lib.rs:198-211 -- no whole-piece shortcut except for single bytes)
__global__ void tk_k_single_front(TkTables T, const uint8_t* __restrict__ text, uint32_t n, TkFrontOut out, int no_lookup) {
    if (blockIdx.x || threadIdx.x) return;
    out.tile_np[0] = 1;
    const uint32_t r = (no_lookup && n > 1u) ? TK_RANK_MAX : tk_lookup_text_piece(T, text, 0, n);
    out.res[TKF_TAIL_NGAP] = 0;
    out.res[TKF_TAIL_NMISS] = r != TK_RANK_MAX ? 0u : 1u;  // (the tail of the one tile's run: what tk_k_count_tiles reads)
    if (r != TK_RANK_MAX) {
        out.res[0] = r;
    } else {  // the one overflow entry of the chunk
        out.res[TKF_TAIL_REFS] = out.data.ovf_base;
        out.res[0] = TK_RES_FLAG | out.data.ovf_base;
        *(uint4*)&out.data.ovf[0].start = make_uint4(0u, n, 0u, 0u);
        out.counters[TK_CNT_OVF] = 1;
        if (n > TK_GLANE_MAX) tk_append_tree(out.listC, out.counters, out.data.ovf_base, 0, n);
    }
}

// ------------------------------------------------------------------------------------------
// Small calls.  Encoding.encode("hello world") is the reference's most common call (tiktoken/core.py:84-139 on one short string): a
// pipeline of twenty launches costs a hundred times what the work does.  One workgroup does the whole job for a single document of up
// to TK_SMALL_MAX bytes without special tokens: text from page-locked host memory straight into LDS, a class per byte, the end of the
// piece that would start at every char (tk_piece_end: the byte-walk scanner, all positions in parallel), one lane follows the chain of
// true piece starts, then the pieces: vocabulary probe, else byte_pair_merge in LDS --
//   * pieces of up to TK_SMALL_PIECE bytes: one lane per piece (tk_lane_merge), 256 at a time;
//   * pieces of up to TK_SMALL_LONG bytes (a URL, an identifier, a word of a script without spaces): sixteen lanes per piece, sixteen
//     pieces at a time, BEFORE the others -- parts as a linked list at the piece's text positions, the leftmost lowest rank by two
//     reductions over the sixteen lanes (lib.rs:151,190), the two new pairs probed by two lanes at the same time;
//   * anything longer that is not a token sends the call to the general path (status 2): a chain of a thousand merges, one after the
//     other, is what tk_k_merge_rounds is for.
// Tokens, their count and the completion word go straight to page-locked host memory, so the host needs neither a copy nor a stream
// synchronisation: it watches the completion word.
// ------------------------------------------------------------------------------------------
#define TK_SMALL_MAX 2048
#define TK_SMALL_PIECE 24
#define TK_SMALL_LONG 256  // (a multiple of 16)
#define TK_SMALL_NO_LONG 0x80000000u  // TkSmallReq::n bit: a piece of more than TK_SMALL_PIECE bytes that is not a token ends the call (status 2)
#define TK_SMALL_HDR 4  // result words before the tokens: status (1 done, 2 not handled), token count, completion sequence number, 0
struct TkSmallAcc {
    const uint8_t *c, *raw;
    uint32_t n;
    __device__ __forceinline__ uint32_t cls(uint64_t pos) const { return pos < n ? (uint32_t)c[pos] : (uint32_t)TK_C_END; }
    __device__ __forceinline__ uint32_t byte(uint64_t pos) const { return pos < n ? (uint32_t)raw[pos] : 0u; }
};
// Up to TK_SMALL_BATCH calls in ONE launch, a workgroup each: callers that arrive together are served by whichever of them gets to launch
// (tk_api.hip, encode_small: the launch path of the HIP runtime is what several threads on one Encoding queue up at).
#define TK_SMALL_BATCH 72
struct TkSmallReq {
    const uint8_t* text;  // the call's text (page-locked, device-visible)
    uint32_t* out;        // its result buffer
    uint32_t* ws;         // merge scratch [256][TK_SMALL_PIECE]
    uint32_t n, seq;
};
struct TkSmallReqs {
    TkSmallReq r[TK_SMALL_BATCH];
};
__global__ __launch_bounds__(256) void tk_k_small(TkTables T, TkSmallReqs R) {
    const uint8_t* __restrict__ text = R.r[blockIdx.x].text;
    uint32_t* __restrict__ out = R.r[blockIdx.x].out;
    uint32_t* __restrict__ ws = R.r[blockIdx.x].ws;
    const uint32_t n = R.r[blockIdx.x].n & ~TK_SMALL_NO_LONG, seq = R.r[blockIdx.x].seq;
    const uint32_t long_max = (R.r[blockIdx.x].n & TK_SMALL_NO_LONG) ? (uint32_t)TK_SMALL_PIECE : (uint32_t)TK_SMALL_LONG;
    __shared__ __attribute__((aligned(16))) uint8_t raw[TK_SMALL_MAX + 16];
    __shared__ uint8_t cls[TK_SMALL_MAX + 16];
    __shared__ uint16_t nxt[TK_SMALL_MAX];
    __shared__ uint16_t plist[TK_SMALL_MAX + 1];
    __shared__ uint32_t lid[TK_SMALL_MAX];  // long pieces: the parts' ids at their text positions; in the end the piece's tokens from its start on
    __shared__ uint16_t llist[TK_SMALL_MAX / (TK_SMALL_PIECE + 1) + 3];  // the long pieces that are not tokens (indices into plist)
    __shared__ uint32_t np_sh, bail_sh, nlong_sh, scan_sh[8];
    // (every merge at the piece's text positions: ids in lid, ranks in lrk -- the one-lane merges as well, so that both kinds run side by side)
    __shared__ uint32_t lrk[TK_SMALL_MAX];
    __shared__ uint16_t lnx[TK_SMALL_MAX], lpv[TK_SMALL_MAX];
    __shared__ uint16_t slist[TK_SMALL_MAX / 2];  // the pieces of 2 .. TK_SMALL_PIECE bytes that are not tokens (indices into plist)
    __shared__ uint32_t nshort_sh;
    (void)ws;
    const uint32_t tid = threadIdx.x;
    const TkPat pat = T.pat;
    for (uint32_t i = tid * 4u; i < TK_SMALL_MAX + 16u; i += 1024u) *(uint32_t*)(raw + i) = i < n ? *(const uint32_t*)(text + i) : 0u;  // (input buffer is padded)
    if (tid == 0) bail_sh = 0, nlong_sh = 0;
    __syncthreads();
    if (tid < 16u && n + tid < TK_SMALL_MAX + 16u) raw[n + tid] = 0;  // (bytes of the last word beyond the text)
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256u) {
        uint32_t c = tk_classify_text(T, raw, i, n);
        if (i == 0) c |= TK_F_HARD;
        cls[i] = (uint8_t)c;
    }
    __syncthreads();
    TkSmallAcc acc{cls, raw, n};
    for (uint32_t i = tid; i < n; i += 256u) {
        uint32_t e = i + 1u;
        if (cls[i] != TK_C_CONT) {
            const uint64_t e64 = tk_piece_end(acc, (uint64_t)i, pat);
            const uint64_t nc = tk_next_char(acc, (uint64_t)i);
            e = (uint32_t)(e64 > nc ? e64 : nc);
            if (e > n) e = n;
        }
        nxt[i] = (uint16_t)e;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t np = 0, p = 0;
        while (p < n) {
            plist[np++] = (uint16_t)p;
            p = nxt[p];
        }
        plist[np] = (uint16_t)n;
        np_sh = np;
    }
    __syncthreads();
    const uint32_t np = np_sh;
    // The one-lane merges and the sixteen-lane merges in ONE phase (written at the end of round 4, first run and shipped in round 5:
    // profiles/r05_small_variants.txt) -- a wavefront takes units, first the long pieces (four to a unit), then the short ones (sixty-four to a unit) --
    // so that the call waits for its longest chain of merges once, not for the long pieces' and then for every round of 256 short ones
    // (profiles/r04_mid_calls_corpus.txt: 2 KiB of web text 250 us against the pipeline's 160).  Every piece's tokens end up in lid from the
    // piece's start on, their number in nxt[start].
    if (tid == 0) nshort_sh = 0;
    __syncthreads();
    for (uint32_t i = tid; i < np; i += 256u) {
        const uint32_t s0 = plist[i], len = (uint32_t)plist[i + 1] - s0;
        const uint32_t tok = tk_lookup_text_piece(T, raw, s0, len);
        if (tok != TK_RANK_MAX) {
            lid[s0] = tok;
            nxt[s0] = 1;
        } else if (len <= TK_SMALL_PIECE) {
            slist[atomicAdd(&nshort_sh, 1u)] = (uint16_t)i;
        } else if (len > long_max) {
            bail_sh = 1;
        } else {
            llist[atomicAdd(&nlong_sh, 1u)] = (uint16_t)i;
        }
    }
    __syncthreads();
    const bool bail = bail_sh != 0;
    const uint32_t nlong = bail ? 0u : nlong_sh, nshort = bail ? 0u : nshort_sh;
    {
        const uint32_t lane = tid & 63u, wid = tid >> 6;
        const uint32_t g = tid & 15u, gsh = tid & 48u;
        const uint32_t ul = (nlong + 3u) >> 2, units = ul + ((nshort + 63u) >> 6);
        for (uint32_t u = wid; u < units; u += 4u) {  // (wave-uniform)
            if (u >= ul) {
                const uint32_t e = (u - ul) * 64u + lane;
                if (e < nshort) {
                    const uint32_t i = slist[e], s0 = plist[i], len = (uint32_t)plist[i + 1] - s0;
                    nxt[s0] = (uint16_t)tk_lane_merge<1>(T, raw, s0, len, lid + s0, lrk + s0, lid + s0);  // (in place: a part moves left or stays)
                }
                continue;
            }
            const uint32_t w = u * 4u + (lane >> 4);
            const bool valid = w < nlong;
            uint32_t s0 = 0, len = 0;
            if (valid) {
                const uint32_t i = llist[w];
                s0 = plist[i];
                len = (uint32_t)plist[i + 1] - s0;
            }
            uint32_t* const id = lid + s0;
            uint32_t* const rk = lrk + s0;
            uint16_t* const nx = lnx + s0;
            uint16_t* const pv = lpv + s0;
            for (uint32_t k = g; k < len; k += 16u) {
                const uint32_t b0 = raw[s0 + k], b1 = raw[s0 + k + 1u];
                id[k] = T.byte_rank[b0];
                rk[k] = k + 1u < len ? T.pair2[(b0 << 8) | b1] : TK_RANK_MAX;
                nx[k] = (uint16_t)(k + 1u);
                pv[k] = (uint16_t)(k ? k - 1u : 0xFFFFu);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (;;) {
                uint32_t br = TK_RANK_MAX, bk = 0xFFFFFFFFu;
                for (uint32_t k = g; k < len; k += 16u) {
                    const uint32_t r = rk[k];
                    if (r < br) br = r, bk = k;
                }
                const uint32_t m = tkm_group_min(br, 4);
                const bool on = m != TK_RANK_MAX;
                if (!__any(on)) break;
                const uint32_t i = tkm_group_min(br == m ? bk : 0xFFFFFFFFu, 4);
                uint32_t j = 0, nn = len, pp = 0xFFFFu, idn = 0, idp = 0;
                if (on) {
                    j = nx[i];
                    nn = nx[j];
                    pp = pv[i];
                    idn = id[nn < len ? nn : i];
                    idp = id[pp != 0xFFFFu ? pp : i];
                }
                const bool right = g == 0u, probe = on && (right ? nn < len : (g == 1u && pp != 0xFFFFu));
                uint32_t newr = TK_RANK_MAX;
                if (probe) newr = tk_probe_pair(T, right ? m : idp, right ? idn : m);
                if (on && g == 0u) {
                    id[i] = m;
                    nx[i] = (uint16_t)nn;
                    if (nn < len) pv[nn] = (uint16_t)i;
                    rk[j] = TK_RANK_MAX;
                    id[j] = TK_RANK_MAX;
                    rk[i] = newr;
                }
                if (on && g == 1u && pp != 0xFFFFu) rk[pp] = newr;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t t = 0;
            for (uint32_t k0 = 0; __any(k0 < len); k0 += 16u) {
                const uint32_t k = k0 + g;
                const uint32_t v = k < len ? id[k] : (uint32_t)TK_RANK_MAX;
                const uint32_t mine = (uint32_t)((__ballot(v != TK_RANK_MAX) >> gsh) & 0xFFFFull);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (v != TK_RANK_MAX) id[t + (uint32_t)__popc(mine & ((1u << g) - 1u))] = v;
                t += (uint32_t)__popc(mine);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            if (valid && g == 0u) nxt[s0] = (uint16_t)t;
        }
    }
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t r0 = 0; r0 < np && !bail; r0 += 256u) {
        const uint32_t i = r0 + tid;
        uint32_t cnt = 0, s0 = 0;
        if (i < np) {
            s0 = plist[i];
            cnt = nxt[s0];
        }
        uint32_t tot;
        const uint32_t ex = tk_block_exscan_256(cnt, &tot, scan_sh);
        uint32_t* o = out + TK_SMALL_HDR + base + ex;
        for (uint32_t j = 0; j < cnt; ++j) o[j] = lid[s0 + j];
        base += tot;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        out[0] = bail_sh ? 2u : 1u;
        out[1] = base;
        __threadfence_system();
        __hip_atomic_store(&out[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
