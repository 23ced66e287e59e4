// Common pieces of the MI355X BPE encode path (gfx950, wave64): constants, wave helpers, the document /
// special-token marking kernels, the scanner's window accessors and the small bitmap -> offsets kernels used by
// the pre-tokenise-only entry point.  The encode pipeline itself is in tk_fused.h.  Included by tk_api.hip only.
//
//   tk_k_mark_docs      document starts -> break bitmap          (core.py:174-176: documents never interact)
//   tk_k_spec_*         (encode() path only) special-token occurrences -> start / interior / break bitmaps
//                                                                 (src/lib.rs:386-402)
//   tk_k_count/_scan_small/_emit   piece-start bitmap -> packed piece offsets (tk_pretokenize_batch)
#pragma once
#include <hip/hip_runtime.h>

#include "tk_device.h"

#define TK_TILE 3840  // text bytes per tile: with 128 bytes of left context and 128 of look-ahead the LDS window is 4096 = 256 lanes x 16

// Deferred pieces are binned by length so that the 64 lanes of a wave run similar trip counts.
#define TK_NBIN 9
#define TK_GLANE_MAX 1024  // longest piece handled by the lane / lane-group kernels; longer ones go to the tree kernel
__host__ __device__ inline uint32_t tk_bin_hi(int b) {
    const uint32_t hi[TK_NBIN] = {16, 24, 32, 48, 64, 128, 256, 512, TK_GLANE_MAX};
    return hi[b];
}
__host__ __device__ inline uint32_t tk_bin_lo(int b) { return b == 0 ? 2u : tk_bin_hi(b - 1) + 1; }
__device__ __forceinline__ int tk_bin_of(uint32_t len) {
    int b = 0;
#pragma unroll
    for (int i = 0; i < TK_NBIN - 1; ++i) b += len > tk_bin_hi(i);
    return b;
}

// counters (device uint32 array)
enum {
    TK_CNT_B = 0, TK_CNT_C = 1, TK_CNT_CBYTES = 2, TK_CNT_CLEVELS = 3, TK_CNT_DUP = 4, TK_CNT_COLL = 5, TK_CNT_ERR = 6, TK_CNT_DEFER = 7,
    TK_CNT_BIN0 = 8,                      // [TK_NBIN] pieces per length bin (tk_k_binfill)
    TK_CNT_RXPOS = 8 + TK_NBIN + 1, TK_CNT_SLOWQ = 8 + TK_NBIN + 2 /* and + 3: the deferred-tile instance's work counters (first list, second list) */, TK_CNT_DEFER2 = 8 + TK_NBIN + 4,
    TK_CNT_OVF = 8 + TK_NBIN + 5,         // overflow entries of the miss data asked for (may exceed the capacity: tk_fused.h, TkMissData)
    TK_CNT_BOFF0 = 8 + TK_NBIN + 6,       // [TK_NBIN] start of each bin's list in listB (tk_k_binfill)
    TK_CNT_N = 8 + 2 * TK_NBIN + 6
};
// (TK_CNT_DEFER2: deferred tiles that gave up their walk -- tk_fused.h, TKF_WALK_BUDGET;
//  TK_CNT_ERR: bits 1, 2 scanner lists of the front kernel; bits 4, 8 the generic pat_str engine -- tk_regex_split.h; TK_CNT_RXPOS: ~position of its first error)

#ifndef TK_MT_BITS
#define TK_MT_BITS 22   // most slots of the in-call miss table (tk_fused.h); sized by the chunk
#endif
#define TK_MT_PROBES 8
#define TK_MAX_LEVELS 6          // 64-ary min-tree levels of the long-piece merge

// ------------------------------------------------------------------------------------------
// wave helpers (wave64)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tk_wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t tk_wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t tk_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t tk_row16_sum(uint32_t v) {  // sum over the aligned 16 lanes of a DPP row, in every lane of it
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]: lane ^ 2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror: the other quad of 8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror: the other half of 16
    return v;
}
// inclusive prefix sum across the wave
// inclusive prefix sum over the wavefront: four DPP row shifts inside the rows of sixteen lanes, then the last lane of a row to the rows
// behind it (row_bcast:15 to rows 1 and 3, row_bcast:31 to rows 2 and 3) -- six full-rate instructions (round 4: six __shfl_up, each a
// ds_bpermute through the LDS crossbar plus a compare and a select)
// (row_bcast:15 / row_bcast:31 are DPP controls of the GFX9 / CDNA encodings only -- the library is built for gfx950; a wave64 target without
// them takes the shuffle form by itself instead of failing in the assembler)
#if !defined(TK_SCAN_SHFL) && defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#define TK_SCAN_SHFL 1
#endif
#ifndef TK_SCAN_SHFL
__device__ __forceinline__ uint32_t tk_wave_scan_u32(uint32_t v, int /*lane*/) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1 (a lane without a source adds nothing)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
#else
__device__ __forceinline__ uint32_t tk_wave_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t w = __shfl_up(v, o, 64);
        if (lane >= o) v += w;
    }
    return v;
}
#endif
// block-wide (256 threads) exclusive scan; returns the exclusive prefix, *total gets the block sum
__device__ __forceinline__ uint32_t tk_block_exscan_256(uint32_t v, uint32_t* total, uint32_t* sh /*[8]*/) {
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = tk_wave_scan_u32(v, lane);
    if (lane == 63) sh[wid] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s = sh[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// wave-aggregated append: every lane with `want` gets a distinct slot index from the LDS counter
__device__ __forceinline__ uint32_t tk_wave_append(bool want, uint32_t* counter, int lane) {
    uint64_t m = __ballot(want);
    if (!m) return 0;
    int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// ------------------------------------------------------------------------------------------
// Start of a chunk: every buffer that has to be zero (or all ones) before the kernels run, in ONE launch (a hipMemsetAsync per buffer
// costs the host 30-60 us each, and with pipelined chunks the host's time per chunk is what bounds the pipeline).
// ------------------------------------------------------------------------------------------
#define TK_CLEAR_MAX 20
struct TkClearArgs {
    uint4* p[TK_CLEAR_MAX];
    uint64_t n16[TK_CLEAR_MAX];  // 16-byte units
    uint32_t v[TK_CLEAR_MAX];    // fill word
    int n;
};
__global__ __launch_bounds__(256) void tk_k_chunk_clear(TkClearArgs a) {
    const uint64_t gtid = blockIdx.x * 256ull + threadIdx.x, gthreads = (uint64_t)gridDim.x * 256;
    for (int r = 0; r < a.n; ++r) {
        const uint4 x = make_uint4(a.v[r], a.v[r], a.v[r], a.v[r]);
        for (uint64_t i = gtid; i < a.n16[r]; i += gthreads) a.p[r][i] = x;
    }
}

// ------------------------------------------------------------------------------------------
// document starts -> bitmaps
// ------------------------------------------------------------------------------------------
// (round 6) ... and, for the back end, the FIRST document that starts in every tile (doc_first[t]; [ntiles]: the first one at or behind the end of
// the text -- empty documents at the end): documents are in the order of their offsets, so a tile's documents are doc_first[t], + 1, ...
// while they start inside it.  tk_k_place writes their token offsets as it passes their pieces (tk_fused.h).
__global__ void tk_k_mark_docs(const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint64_t base, uint64_t n,
                               uint32_t* __restrict__ brk, uint32_t* __restrict__ docb, uint32_t* __restrict__ doc_first, uint64_t ntiles) {
    for (uint64_t d = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; d < n_docs; d += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t pos = doc_off[d] - base;
        if (pos < n) {
            atomicOr(&brk[pos >> 5], 1u << (pos & 31));
            if (docb) atomicOr(&docb[pos >> 5], 1u << (pos & 31));
        }
        if (doc_first) atomicMin(&doc_first[pos < n ? pos / TK_TILE : ntiles], (uint32_t)d);
    }
}

// ------------------------------------------------------------------------------------------
// special tokens (encode() with allowed_special; src/lib.rs:386-402, 426-434)
// ------------------------------------------------------------------------------------------
// Longest allowed special token that matches at text[pos..] without crossing a document start.
// Returns its length (0 = none) and index.  The next 32 text bytes and the document starts among the next 64 positions are read once; a
// special token of at most 32 bytes is compared with them word by word (four independent loads of its bytes instead of a load per byte
// that waits for the byte before it: the resolving pass calls this two or three times per candidate, one lane of a wavefront at a time).
__device__ __forceinline__ uint64_t tk_bits64(const uint32_t* __restrict__ bm, uint64_t pos) {  // bits [pos, pos + 64) of a bitmap (readable two words past them)
    const uint64_t wi = pos >> 5;
    const uint32_t sh = (uint32_t)(pos & 31);
    const uint32_t w0 = bm[wi], w1 = bm[wi + 1], w2 = bm[wi + 2];
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    return sh ? ((lo >> sh) | ((uint64_t)w2 << (64u - sh))) : lo;
}
__device__ __forceinline__ uint32_t tk_special_at(const TkTables& T, const uint8_t* __restrict__ text, uint64_t pos, uint64_t n,
                                                  const uint8_t* __restrict__ allowed, const uint32_t* __restrict__ docb,
                                                  uint32_t* idx_out) {
    uint32_t b0 = text[pos];
    if (!((T.spec_first[b0 >> 5] >> (b0 & 31)) & 1u)) return 0;
    uint64_t tw[4];  // (the text is readable 64 bytes past n)
#pragma unroll
    for (int i = 0; i < 4; ++i) tw[i] = tk_load8(text, pos + 8u * i);
    const uint64_t db = docb ? tk_bits64(docb, pos + 1) : 0ull;  // document starts at pos + 1 .. pos + 64
    uint32_t best = 0, bi = 0;
    for (uint32_t k = 0; k < T.n_spec; ++k) {
        // (round 6: a token's first eight bytes, length and offset in one load that depends on nothing -- the loads of all tokens are in flight
        // together; a candidate that is no special token, "<|x", leaves after it.  Offsets, first byte and bytes were four dependent loads per token.)
        const uint4 hd = ((const uint4*)T.spec_head)[k];
        const uint32_t o = hd.w, len = hd.z;
        if (len <= best || pos + len > n) continue;
        if (tk_mask_low_bytes(tw[0] ^ (((uint64_t)hd.y << 32) | hd.x), len < 8u ? len : 8u) != 0ull) continue;
        if (allowed && !allowed[k]) continue;
        bool ok = true;
        if (len <= 32u) {
#pragma unroll
            for (int i = 1; i < 4; ++i)
                if (8u * i < len) ok = ok && tk_mask_low_bytes(tw[i] ^ tk_load8(T.spec_bytes, (uint64_t)o + 8u * i), len - 8u * i) == 0ull;
            ok = ok && (db & ((1ull << (len - 1u)) - 1ull)) == 0ull;
        } else {
            for (uint32_t i = 1; i < len && ok; ++i) ok = (text[pos + i] == T.spec_bytes[o + i]) && !(docb && tk_bit(docb, pos + i));
        }
        if (ok) {
            best = len;
            bi = k;
        }
    }
    *idx_out = bi;
    return best;
}

// Candidates: positions at which an allowed special token matches.  16 text bytes per thread; almost every byte fails the first-byte
// test, so the kernel is one coalesced read of the text.  With at most four distinct first bytes (every stock encoding: '<') the test is
// four byte-equality tests per 32-bit word (x ^ c has a zero byte; the borrow may mark a byte above a true hit as well: a false
// candidate, which tk_special_at rejects); otherwise the 256-bit set decides byte by byte.
__global__ void tk_k_spec_cand(TkTables T, const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ allowed,
                               const uint32_t* __restrict__ docb, uint32_t* __restrict__ cand) {
    for (uint64_t p0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 16; p0 < n; p0 += (uint64_t)gridDim.x * blockDim.x * 16) {
        const uint4 v = *(const uint4*)(text + p0);  // (text is readable 64 bytes past n)
        const uint32_t nxt = *(const uint32_t*)(text + p0 + 16) & 0xFFu;  // (the byte behind the sixteen, asked for together with them: inside `if (hits)` it was a second trip for every wavefront with a '<')
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t hits = 0;
        if (T.n_spec_fb <= 4u) {
            for (uint32_t f = 0; f < T.n_spec_fb; ++f) {
                const uint32_t c4 = ((T.spec_fb >> (8u * f)) & 0xFFu) * 0x01010101u;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t x = w[d] ^ c4;
                    const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;  // bit 7 of every zero byte (and, rarely, of a 0x01 above one)
                    // bits 7, 15, 23, 31 -> bits 0..3
                    hits |= (((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu) << (4 * d);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t b = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                hits |= ((T.spec_first[b >> 5] >> (b & 31)) & 1u) << k;
            }
        }
        if (hits) {  // the byte behind a hit must be some special token's second byte ("<" is common in web text, "<|" is not)
            uint32_t keep = 0;
            for (uint32_t m = hits; m; m &= m - 1) {
                const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
                const uint32_t b1 = k < 15u ? (w[(k + 1) >> 2] >> (((k + 1) & 3u) * 8u)) & 0xFFu : nxt;
                keep |= ((T.spec_second[b1 >> 5] >> (b1 & 31u)) & 1u) << k;
            }
            hits = keep;
        }
        while (hits) {
            const uint64_t pos = p0 + (uint32_t)(__ffs((int)hits) - 1);
            hits &= hits - 1;
            uint32_t idx;
            if (pos < n && tk_special_at(T, text, pos, n, allowed, docb, &idx)) atomicOr(&cand[pos >> 5], 1u << (pos & 31));
        }
    }
}

// Resolve overlapping candidates exactly as a left-to-right search would (src/lib.rs:389-401,432):
// a candidate is taken iff the greedy non-overlapping walk from the head of its overlap cluster
// lands on it.  (Earlier candidates are looked for in the 64 bits of the bitmap before the position -- two loads, not one per bit;
// special tokens of 66 bytes and more take the walk bit by bit.)
__global__ void tk_k_spec_resolve(TkTables T, const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ allowed,
                                  const uint32_t* __restrict__ docb, const uint32_t* __restrict__ cand, uint32_t max_len,
                                  uint32_t* __restrict__ spec_start, uint32_t* __restrict__ spec_in, uint32_t* __restrict__ brk) {
    // one thread per 32-position word of the candidate bitmap (candidates are sparse)
    const uint64_t nwords = (n + 31) / 32;
    for (uint64_t wi = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; wi < nwords; wi += (uint64_t)gridDim.x * blockDim.x)
      for (uint32_t cw = cand[wi]; cw; cw &= cw - 1) {
        const uint64_t pos = wi * 32 + (uint32_t)(__ffs((int)cw) - 1);
        uint32_t idx;
        // head of the overlap cluster: walk left while some earlier candidate's span covers `h`
        uint64_t h = pos;
        for (;;) {
            bool moved = false;
            uint64_t lo = h >= (uint64_t)(max_len - 1) ? h - (max_len - 1) : 0;
            if (max_len <= 65u && h >= 64u) {
                uint64_t before = tk_bits64(cand, h - 64u);          // candidates at h - 64 .. h - 1 (bit 63 = h - 1)
                before = h > lo ? (before & (~0ull << (64u - (uint32_t)(h - lo)))) : 0ull;  // only those at lo .. h - 1
                while (before) {  // nearest first
                    const uint32_t b = 63u - (uint32_t)__clzll((long long)before);
                    before &= ~(1ull << b);
                    const uint64_t j = h - 64u + b;
                    if (j + tk_special_at(T, text, j, n, allowed, docb, &idx) > h) {
                        h = j;
                        moved = true;
                        break;
                    }
                }
            } else {
                for (uint64_t j = h; j-- > lo;) {
                    if (tk_bit(cand, j) && j + tk_special_at(T, text, j, n, allowed, docb, &idx) > h) {
                        h = j;
                        moved = true;
                        break;
                    }
                }
            }
            if (!moved) break;
        }
        // greedy walk from the head
        uint64_t cur = h;
        bool taken = false;
        while (cur <= pos) {
            if (cur == pos) {
                taken = true;
                break;
            }
            uint64_t nx = cur + tk_special_at(T, text, cur, n, allowed, docb, &idx);
            while (nx <= pos && !tk_bit(cand, nx)) ++nx;
            cur = nx;
        }
        if (!taken) continue;
        uint32_t len = tk_special_at(T, text, pos, n, allowed, docb, &idx);
        atomicOr(&spec_start[pos >> 5], 1u << (pos & 31));
        atomicOr(&brk[pos >> 5], 1u << (pos & 31));
        for (uint64_t a = pos + 1, b = pos + len; a < b;) {  // the token's interior, a word of the bitmap at a time
            const uint64_t we = (a | 31u) + 1u, hi = we < b ? we : b;
            const uint32_t cnt = (uint32_t)(hi - a);
            atomicOr(&spec_in[a >> 5], (cnt >= 32u ? 0xFFFFFFFFu : ((1u << cnt) - 1u)) << (a & 31u));
            a = hi;
        }
        if (pos + len < n) atomicOr(&brk[(pos + len) >> 5], 1u << ((pos + len) & 31));
    }
}

// id of the special token whose bytes are text[pos..pos+len)
__device__ __forceinline__ uint32_t tk_special_id(const TkTables& T, const uint8_t* __restrict__ text, uint64_t pos, uint32_t len) {
    for (uint32_t k = 0; k < T.n_spec; ++k) {
        uint32_t o = T.spec_off[k];
        if (T.spec_off[k + 1] - o != len) continue;
        bool ok = true;
        for (uint32_t i = 0; i < len && ok; ++i) ok = text[pos + i] == T.spec_bytes[o + i];
        if (ok) return T.spec_id[k];
    }
    return TK_RANK_MAX;
}

// ------------------------------------------------------------------------------------------
// pre-tokenisation: tile geometry and the scanner's accessors (the kernel is tk_k_front in tk_fused.h)
// A tile is 3840 bytes; its LDS window adds 128 bytes of left context and 128 of look-ahead: 4096 bytes = 16 per lane.  (Left context:
// where the last certain start before the tile is looked for -- 2.0 % of the tiles of the bench corpus have none within 64 bytes, 0.8 %
// none within 128; look-ahead: a piece that reaches further beyond the tile end is finished by the workgroup-wide scanner.)
// ------------------------------------------------------------------------------------------
#define TK2_LEFT 128
#define TK2_RIGHT 128
#define TK2_WIN (TK2_LEFT + TK_TILE + TK2_RIGHT)  // 4096
#define TK2_NSEG (TK2_WIN / 64)                   // 64
#define TK2_CLIST 2048                            // certain-start list entries per tile (more: every lane scans from its own starts)

// Out-of-line slow paths: they are rare, and inlining them at every call site of the scanner made the
// kernel ~30k instructions (instruction-cache thrash).
__device__ __noinline__ uint32_t tk_class_byte_slow(const TkTables* T, const uint8_t* text, uint64_t pos, uint64_t n, const uint32_t* brk,
                                                    const uint32_t* ss, const uint32_t* si) {
    return tk_class_byte(*T, text, pos, n, brk, ss, si);
}

// class nibble of window position r from the four plane bitmaps in LDS (32-bit view: plane p starts at word p * PLW)
#define TK2_PLW (2 * (TK2_NSEG + 2))
__device__ __forceinline__ uint32_t tk_class_at_lds(const uint32_t* planes32, uint32_t r) {
    const uint32_t wi = r >> 5, b = r & 31u;
    return ((planes32[wi] >> b) & 1u) | (((planes32[TK2_PLW + wi] >> b) & 1u) << 1) | (((planes32[2 * TK2_PLW + wi] >> b) & 1u) << 2) |
           (((planes32[3 * TK2_PLW + wi] >> b) & 1u) << 3);
}

struct TkWin2Acc {  // byte-walking fallback inside the window (for the pieces the bit-parallel scanners decline).  It never leaves
                    // the window: a look outside sets `left` and reads as end-of-text -- the caller then hands the piece to the
                    // workgroup-wide scanner (tk_coop_*), which answers run queries at 4 KiB per step.
    const uint32_t* planes32;  // [4][TK2_PLW]
    const uint32_t* start32;   // char-start bitmap
    const uint32_t* hard32;    // hard-start bitmap
    const uint8_t* raw;
    int64_t base;
    uint64_t n;
    bool left;
    __device__ __forceinline__ uint32_t cls(uint64_t pos) {
        if (pos >= n) return TK_C_END;
        int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < TK2_WIN) {
            const uint32_t wi = (uint32_t)r >> 5, b = (uint32_t)r & 31u;
            if (!((start32[wi] >> b) & 1u)) return (uint32_t)TK_C_CONT;
            return tk_class_at_lds(planes32, (uint32_t)r) | (((hard32[wi] >> b) & 1u) << 7);
        }
        left = true;
        return TK_C_END;
    }
    __device__ __forceinline__ uint32_t byte(uint64_t pos) {
        int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < TK2_WIN) return raw[r];
        left = true;
        return 0;
    }
};

struct TkBmExt {  // extension windows for runs longer than the first 64-bit window (LDS bitmaps of the tile)
    const uint64_t (*bm)[TK2_NSEG + 2];
    uint32_t wi, sh, lim;
    __device__ __forceinline__ uint64_t win(int kind, uint32_t j) const {
        const uint32_t w0 = wi + j;
        if (w0 + 1 > TK2_NSEG + 1) return kind == TKB_HARD ? ~0ull : 0ull;
        return sh ? ((bm[kind][w0] >> sh) | (bm[kind][w0 + 1] << (64u - sh))) : bm[kind][w0];
    }
    __device__ __forceinline__ uint32_t limit() const { return lim; }
};

__device__ __forceinline__ uint64_t tk_piece_end_slow(TkWin2Acc* acc, uint64_t p, TkPat pat) {
    uint64_t e = tk_piece_end(*acc, p, pat);
    if (e <= p) e = tk_next_char(*acc, p);
    return e;
}
// ------------------------------------------------------------------------------------------
// bitmap -> piece offsets
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tk_k_count(const uint32_t* __restrict__ starts, uint64_t nwords, uint32_t* __restrict__ blockcnt) {
    __shared__ uint32_t sh[8];
    uint64_t w = blockIdx.x * 256ull + threadIdx.x;
    uint32_t c = w < nwords ? __popc(starts[w]) : 0;
    c = tk_wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// The scan kernels run with workgroups of 256 threads: in a multi-chunk batch they are queued beside the next chunk's front kernel, whose
// workgroups (256 threads, 64 registers) fill the device -- a workgroup of the same shape gets the place of one that ends, one of 1024
// threads waits until the front kernel is over (seen in round 4: 2.3 ms for a 20 us scan).
#define TK_SCAN_THREADS 256
// single-workgroup exclusive scan of a (small) uint32 array in place; total -> total_out[0] (u64)
__global__ __launch_bounds__(TK_SCAN_THREADS) void tk_k_scan_small(uint32_t* __restrict__ a, uint64_t n, uint64_t* __restrict__ total_out) {
    constexpr int NW = TK_SCAN_THREADS / 64;
    __shared__ uint32_t wsum[NW];
    __shared__ uint64_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int E = 16;  // consecutive elements per thread
    for (uint64_t base = 0; base < n; base += TK_SCAN_THREADS * E) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * E;
        uint32_t v[E], mine = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            v[j] = i0 + j < n ? a[i0 + j] : 0;
            mine += v[j];
        }
        const uint32_t inc = tk_wave_scan_u32(mine, lane);
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (w < wid) wbase += wsum[w];
            tot += wsum[w];
        }
        const uint64_t carry = carry_sh;
        uint32_t run = (uint32_t)(carry + wbase + inc - mine);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (i0 + j < n) a[i0 + j] = run;
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_sh = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total_out[0] = carry_sh;
}

// Larger arrays: reduce, scan the sums, scan again with the bases (three short launches instead of one workgroup walking everything).
#define TK_SCAN_BLOCK 4096  // elements per workgroup of the two wide passes (256 threads x 16)
__global__ __launch_bounds__(TK_SCAN_THREADS) void tk_k_scan_sums(const uint32_t* __restrict__ a, uint64_t n, uint32_t* __restrict__ sums) {
    constexpr int NW = TK_SCAN_THREADS / 64, E = TK_SCAN_BLOCK / TK_SCAN_THREADS;
    __shared__ uint32_t wsum[NW];
    const uint64_t i0 = (uint64_t)blockIdx.x * TK_SCAN_BLOCK + (uint64_t)threadIdx.x * E;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) mine += i0 + j < n ? a[i0 + j] : 0u;
    mine = tk_wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += wsum[w];
        sums[blockIdx.x] = t;
    }
}
// bases[b] = exclusive prefix of the sums (tk_k_scan_small over them); a <- exclusive prefix of a, in place
__global__ __launch_bounds__(TK_SCAN_THREADS) void tk_k_scan_apply(uint32_t* __restrict__ a, uint64_t n, const uint32_t* __restrict__ bases) {
    constexpr int NW = TK_SCAN_THREADS / 64, E = TK_SCAN_BLOCK / TK_SCAN_THREADS;
    __shared__ uint32_t wsum[NW];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * TK_SCAN_BLOCK + (uint64_t)threadIdx.x * E;
    uint32_t v[E], mine = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        v[j] = i0 + j < n ? a[i0 + j] : 0u;
        mine += v[j];
    }
    const uint32_t inc = tk_wave_scan_u32(mine, lane);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wid; ++w) wbase += wsum[w];
    uint32_t run = bases[blockIdx.x] + wbase + inc - mine;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (i0 + j < n) a[i0 + j] = run;
        run += v[j];
    }
}

__global__ __launch_bounds__(256) void tk_k_emit(const uint32_t* __restrict__ starts, uint64_t nwords, const uint32_t* __restrict__ blockpre,
                                                 uint32_t* __restrict__ pstart, uint64_t P, uint64_t n, const uint32_t* __restrict__ gapb) {
    __shared__ uint32_t sh[8];
    uint64_t w = blockIdx.x * 256ull + threadIdx.x;
    uint32_t v = w < nwords ? starts[w] : 0;
    uint32_t tot;
    uint32_t ex = tk_block_exscan_256(__popc(v), &tot, sh);
    uint64_t o = (uint64_t)blockpre[blockIdx.x] + ex;
    while (v) {
        uint32_t b = __ffs(v) - 1;
        v &= v - 1;
        // (bit 31: a gap char of the generic engine's split -- tk_pretokenize_batch handles one chunk, positions stay below 2^30)
        pstart[o++] = (uint32_t)(w * 32 + b) | ((gapb && ((gapb[w] >> b) & 1u)) ? 0x80000000u : 0u);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pstart[P] = (uint32_t)n;
}
