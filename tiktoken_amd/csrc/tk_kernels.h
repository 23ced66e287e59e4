// HIP kernels of the MI355X BPE encode path (gfx950, wave64).  Included by tk_api.hip only.
//
// Pipeline (one launch sequence per chunk of packed documents, all intermediates in HBM):
//
//   tk_k_mark_docs      document starts -> break bitmap                      (core.py:174-176: documents
//                                                                              never interact)
//   tk_k_spec_*         (encode() path only) special-token occurrences -> start / interior / break
//                       bitmaps                                               (src/lib.rs:386-402)
//   tk_k_pretok2<PAT>   regex pre-tokenisation -> piece-start bitmap          (src/lib.rs:365); bit-parallel
//                       (tk_k_pretok is the byte-walking original, kept behind TIKTOKEN_AMD_DEBUG=32)
//   tk_k_count/_scan/_emit   bitmap -> packed piece offsets
//   tk_k_lookup         whole-piece probe; misses are de-duplicated through a miss table and the
//                       distinct ones appended to length-binned lists         (src/lib.rs:367-369)
//   tk_k_merge_llane<N> one LANE per 2..64-byte piece: byte_pair_merge in LDS  (src/lib.rs:140-196)
//   tk_k_merge_group<G> G lanes per 65..1024-byte piece
//   tk_k_merge_long     one wavefront per longer piece, 64-ary min tree in HBM scratch
//                                                                              (same result as lib.rs:47-138)
//   tk_k_dup_fix        duplicates: byte-verify against the claimant, copy its result
//   tk_k_scan_*         token counts -> token offsets
//   tk_k_gather         tokens into their final packed order; tk_k_docoff per-document offsets
#pragma once
#include <hip/hip_runtime.h>

#include "tk_device.h"

#define TK_TILE 4096
#define TK_HALO_L 4
#define TK_HALO_R 252
#define TK_WIN (TK_HALO_L + TK_TILE + TK_HALO_R)
#define TK_LANE_MAX 16   // longest piece merged by a single lane inside tk_k_lookup (LDS scratch)
#define TK_PPT 4         // pieces per thread per block iteration in tk_k_lookup

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
struct TkBins {
    uint32_t off[TK_NBIN];  // start of each bin's list inside the pool
};

// counters (device uint32 array)
enum { TK_CNT_B = 0, TK_CNT_C = 1, TK_CNT_CBYTES = 2, TK_CNT_CLEVELS = 3, TK_CNT_DUP = 4, TK_CNT_COLL = 5, TK_CNT_BIN0 = 8, TK_CNT_N = 8 + TK_NBIN + 1 };

// In-call de-duplication of missed pieces (the same rare word occurs many times in a batch): a
// best-effort open-addressed table {hash(bytes, len) -> first piece that claimed it}.  The claimant is
// merged; later identical pieces are verified byte for byte against it (tk_k_dup_fix) and copy its
// result.  Nothing is carried over between calls.
#define TK_MT_BITS 22
#define TK_MT_PROBES 8
struct TkMissTable {
    unsigned long long* key;  // [1 << TK_MT_BITS], ~0 = empty
    uint32_t* rep;            // claimant piece index
};

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
// inclusive prefix sum across the wave
__device__ __forceinline__ uint32_t tk_wave_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t w = __shfl_up(v, o, 64);
        if (lane >= o) v += w;
    }
    return v;
}
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
// document starts -> bitmaps
// ------------------------------------------------------------------------------------------
__global__ void tk_k_mark_docs(const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint64_t base, uint64_t n,
                               uint32_t* __restrict__ brk, uint32_t* __restrict__ docb) {
    for (uint64_t d = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; d < n_docs; d += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t pos = doc_off[d] - base;
        if (pos < n) {
            atomicOr(&brk[pos >> 5], 1u << (pos & 31));
            if (docb) atomicOr(&docb[pos >> 5], 1u << (pos & 31));
        }
    }
}

// ------------------------------------------------------------------------------------------
// special tokens (encode() with allowed_special; src/lib.rs:386-402, 426-434)
// ------------------------------------------------------------------------------------------
// Longest allowed special token that matches at text[pos..] without crossing a document start.
// Returns its length (0 = none) and index.
__device__ __forceinline__ uint32_t tk_special_at(const TkTables& T, const uint8_t* __restrict__ text, uint64_t pos, uint64_t n,
                                                  const uint8_t* __restrict__ allowed, const uint32_t* __restrict__ docb,
                                                  uint32_t* idx_out) {
    uint32_t b0 = text[pos];
    if (!((T.spec_first[b0 >> 5] >> (b0 & 31)) & 1u)) return 0;
    uint32_t best = 0, bi = 0;
    for (uint32_t k = 0; k < T.n_spec; ++k) {
        if (allowed && !allowed[k]) continue;
        uint32_t o = T.spec_off[k], len = T.spec_off[k + 1] - o;
        if (len <= best || pos + len > n || T.spec_bytes[o] != b0) continue;
        bool ok = true;
        for (uint32_t i = 1; i < len && ok; ++i) ok = (text[pos + i] == T.spec_bytes[o + i]) && !(docb && tk_bit(docb, pos + i));
        if (ok) {
            best = len;
            bi = k;
        }
    }
    *idx_out = bi;
    return best;
}

__global__ void tk_k_spec_cand(TkTables T, const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ allowed,
                               const uint32_t* __restrict__ docb, uint32_t* __restrict__ cand) {
    for (uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; pos < n; pos += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t idx;
        if (tk_special_at(T, text, pos, n, allowed, docb, &idx)) atomicOr(&cand[pos >> 5], 1u << (pos & 31));
    }
}

// Resolve overlapping candidates exactly as a left-to-right search would (src/lib.rs:389-401,432):
// a candidate is taken iff the greedy non-overlapping walk from the head of its overlap cluster
// lands on it.
__global__ void tk_k_spec_resolve(TkTables T, const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ allowed,
                                  const uint32_t* __restrict__ docb, const uint32_t* __restrict__ cand, uint32_t max_len,
                                  uint32_t* __restrict__ spec_start, uint32_t* __restrict__ spec_in, uint32_t* __restrict__ brk) {
    for (uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; pos < n; pos += (uint64_t)gridDim.x * blockDim.x) {
        if (!tk_bit(cand, pos)) continue;
        uint32_t idx;
        // head of the overlap cluster: walk left while some earlier candidate's span covers `h`
        uint64_t h = pos;
        for (;;) {
            bool moved = false;
            uint64_t lo = h >= (uint64_t)(max_len - 1) ? h - (max_len - 1) : 0;
            for (uint64_t j = h; j-- > lo;) {
                if (tk_bit(cand, j) && j + tk_special_at(T, text, j, n, allowed, docb, &idx) > h) {
                    h = j;
                    moved = true;
                    break;
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
        for (uint64_t j = pos + 1; j < pos + len; ++j) atomicOr(&spec_in[j >> 5], 1u << (j & 31));
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
// pre-tokenisation
// ------------------------------------------------------------------------------------------
struct TkWinAcc {
    const uint8_t* win;  // LDS window of class bytes
    int64_t base;        // text position of win[0]
    const TkTables* T;
    const uint8_t* text;
    uint64_t n;
    const uint32_t *brk, *ss, *si;
    __device__ __forceinline__ uint32_t cls(uint64_t pos) const {
        if (pos >= n) return TK_C_END;
        int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < TK_WIN) return win[r];
        return tk_class_byte(*T, text, pos, n, brk, ss, si);  // beyond the window: classify from HBM
    }
    __device__ __forceinline__ uint32_t byte(uint64_t pos) const { return text[pos]; }
};

// One workgroup per 4 KiB tile.  Class bytes of the tile (+ a 4-byte left and 252-byte right halo)
// are staged in LDS; every char that is a *certain* piece start (tk_certain_start) runs the
// sequential scanner from there until it reaches the next certain start, marking the uncertain
// boundaries it finds on the way.  Certain starts are dense in real text (every word), so a lane
// typically evaluates one piece.  Units never communicate: the rule only needs the previous char.
__global__ __launch_bounds__(256) void tk_k_pretok(TkTables T, const uint8_t* __restrict__ text, uint64_t n,
                                                   const uint32_t* __restrict__ brk, const uint32_t* __restrict__ ss,
                                                   const uint32_t* __restrict__ si, uint32_t* __restrict__ starts, int dbg) {
    __shared__ uint8_t win[TK_WIN];
    __shared__ uint32_t bits[TK_TILE / 32];
    const uint64_t tile_start = (uint64_t)blockIdx.x * TK_TILE;
    const int64_t base = (int64_t)tile_start - TK_HALO_L;
    for (int w = threadIdx.x; w < TK_WIN; w += 256) {
        int64_t gp = base + w;
        uint32_t c = TK_C_END;
        if (gp >= 0 && (uint64_t)gp < n) c = tk_class_byte(T, text, (uint64_t)gp, n, brk, ss, si);
        win[w] = (uint8_t)c;
    }
    if (threadIdx.x < TK_TILE / 32) bits[threadIdx.x] = 0;
    __syncthreads();
    TkWinAcc acc{win, base, &T, text, n, brk, ss, si};
    const int pat = T.pattern;
    for (int k = 0; k < TK_TILE / 256; ++k) {
        uint32_t il = threadIdx.x + k * 256;
        uint64_t gp = tile_start + il;
        if (gp >= n) break;
        uint32_t c = win[TK_HALO_L + il];
        if ((c & 15u) == TK_C_CONT) continue;
        bool certain = (c & TK_F_HARD) != 0;
        if (!certain) {
            int j = (int)(TK_HALO_L + il) - 1;
            while (j > 0 && win[j] == TK_C_CONT) --j;
            certain = tk_certain_start(pat, win[j] & 15u, c & 15u);
        }
        if (!certain) continue;
        atomicOr(&bits[il >> 5], 1u << (il & 31));
        uint64_t p = gp;
        if (dbg & 16) continue;
        for (;;) {
            uint64_t e = tk_piece_end(acc, p, pat);
            if (e <= p) e = tk_next_char(acc, p);  // defensive; cannot happen
            if (e >= n) break;
            uint32_t ce = acc.cls(e);
            if (ce & TK_F_HARD) break;
            uint64_t j = e - 1;
            while (acc.cls(j) == TK_C_CONT) --j;
            if (tk_certain_start(pat, acc.cls(j) & 15u, ce & 15u)) break;
            if (e < tile_start + TK_TILE)
                atomicOr(&bits[(uint32_t)(e - tile_start) >> 5], 1u << ((uint32_t)(e - tile_start) & 31));
            else
                atomicOr(&starts[e >> 5], 1u << (e & 31));
            p = e;
        }
    }
    __syncthreads();
    if (threadIdx.x < TK_TILE / 32) {
        uint32_t v = bits[threadIdx.x];
        uint64_t wi = tile_start / 32 + threadIdx.x;
        if (v) atomicOr(&starts[wi], v);
    }
}

// ------------------------------------------------------------------------------------------
// pre-tokenisation, bit-parallel (the production kernel; tk_k_pretok above is the byte-walking
// original, kept as the reference implementation behind TIKTOKEN_AMD_DEBUG=32)
//
// One workgroup per 4 KiB tile; window = 64 B left context + tile + 192 B right halo (68 wave-sized
// segments).  Phase A: 16-byte vector loads of the window into LDS.  Phase B: one wave per 64-byte
// segment, lane = byte: class of every byte (continuation bytes inherit their char's class), then
// eleven `__ballot`s give one 64-bit word of each class bitmap.  Phase C: certain piece starts
// (previous byte's class x this class) are compacted into an LDS list.  Phase D: one lane per certain
// start evaluates pieces with tk_piece_len_bits -- run ends are `ctz` on 64-bit bitmap windows --
// until the next certain start.  Phase E: the tile's 512-byte slice of the start bitmap is OR-ed out.
// ------------------------------------------------------------------------------------------
#define TK2_LEFT 64
#define TK2_RIGHT 192
#define TK2_WIN (TK2_LEFT + TK_TILE + TK2_RIGHT)  // 4352
#define TK2_NSEG (TK2_WIN / 64)                   // 68
#define TK2_CLIST 1536

// Out-of-line slow paths: they are rare, and inlining them at every call site of the scanner made the
// kernel ~30k instructions (instruction-cache thrash).
__device__ __noinline__ uint32_t tk_class_byte_slow(const TkTables* T, const uint8_t* text, uint64_t pos, uint64_t n, const uint32_t* brk,
                                                    const uint32_t* ss, const uint32_t* si) {
    return tk_class_byte(*T, text, pos, n, brk, ss, si);
}

struct TkWin2Acc {  // byte-walking fallback: propagated classes inside the window, HBM outside
    const uint8_t* cls2;
    const uint8_t* raw;
    int64_t base;
    const TkTables* T;
    const uint8_t* text;
    uint64_t n;
    const uint32_t *brk, *ss, *si;
    __device__ __forceinline__ uint32_t cls(uint64_t pos) const {
        if (pos >= n) return TK_C_END;
        int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < TK2_WIN) {
            uint32_t c = cls2[r];
            return (c & 0x40u) ? (uint32_t)TK_C_CONT : (c & 0x8Fu);
        }
        return tk_class_byte_slow(T, text, pos, n, brk, ss, si);
    }
    __device__ __forceinline__ uint32_t byte(uint64_t pos) const {
        int64_t r = (int64_t)pos - base;
        if (r >= 0 && r < TK2_WIN) return raw[r];
        return text[pos];
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

__device__ __forceinline__ uint64_t tk_piece_end_slow(TkWin2Acc* acc, uint64_t p, int pat) {
    uint64_t e = tk_piece_end(*acc, p, pat);
    if (e <= p) e = tk_next_char(*acc, p);
    return e;
}
// class byte of e (flags included) and class of the char before e, for positions outside the LDS window
__device__ __forceinline__ uint32_t tk_boundary_classes_slow(const TkWin2Acc* acc, uint64_t e) {
    uint32_t ce = acc->cls(e);
    uint64_t j = e - 1;
    while (acc->cls(j) == TK_C_CONT) --j;
    return (ce & 0xFFu) | ((acc->cls(j) & 15u) << 8);
}

template <int PAT>
__global__ __launch_bounds__(256) void tk_k_pretok2(TkTables T, const uint8_t* __restrict__ text, uint64_t n,
                                                    const uint32_t* __restrict__ brk, const uint32_t* __restrict__ ss,
                                                    const uint32_t* __restrict__ si, uint32_t* __restrict__ starts,
                                                    unsigned long long* __restrict__ prof) {
#define TK_PROF(slot)                                                             \
    if (prof && threadIdx.x == 0) {                                               \
        long long t__ = __builtin_readcyclecounter();                             \
        atomicAdd(&prof[slot], (unsigned long long)(t__ - t_prev));               \
        t_prev = t__;                                                             \
    }
    long long t_prev = prof ? __builtin_readcyclecounter() : 0;
    __shared__ __attribute__((aligned(16))) uint8_t raw[TK2_WIN + 16];
    __shared__ uint8_t cls2[TK2_WIN];
    __shared__ uint64_t bm[TKB_KINDS][TK2_NSEG + 2];
    __shared__ uint32_t bits[TK_TILE / 32];
    __shared__ uint16_t clist[TK2_CLIST];  // certain starts of the tile (overflow handled in place)
    __shared__ uint32_t cn;
    __shared__ uint32_t certm[16];  // certain-start masks of this pattern, by class of the previous char
    __shared__ uint32_t brkw[TK2_WIN / 32 + 1], ssw[TK2_WIN / 32 + 1], siw[TK2_WIN / 32 + 1];
    __shared__ __attribute__((aligned(16))) uint8_t st1[0x1100];
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_start = (uint64_t)blockIdx.x * TK_TILE;
    const int64_t base = (int64_t)tile_start - TK2_LEFT;
    // ---- A: window -> LDS
    for (uint32_t v = tid; v < (TK2_WIN + 16) / 16; v += 256) {
        int64_t gp = base + (int64_t)v * 16;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (gp >= 0 && (uint64_t)gp < n) x = *(const uint4*)(text + gp);  // text is readable 64 bytes past n
        *(uint4*)(raw + v * 16) = x;
    }
    for (uint32_t v = tid; v < 0x1100 / 16; v += 256) *(uint4*)(st1 + v * 16) = *(const uint4*)(T.uc_stage1 + v * 16);
    if (tid < TK2_WIN / 32) {  // break / special bitmaps of the window (the window base is 32-aligned)
        int64_t wgp = base + (int64_t)tid * 32;
        bool in = wgp >= 0 && (uint64_t)wgp < n;
        brkw[tid] = in ? brk[wgp >> 5] : 0u;
        ssw[tid] = (in && ss) ? ss[wgp >> 5] : 0u;
        siw[tid] = (in && si) ? si[wgp >> 5] : 0u;
    }
    if (tid < 16) certm[tid] = tk_certain_mask(PAT, tid);
    if (tid == 0) cn = 0;
    if (tid < TKB_KINDS) {
        bm[tid][TK2_NSEG] = tid <= TKB_HARD ? ~0ull : 0ull;  // beyond the window: unknown -> "stop"
        bm[tid][TK2_NSEG + 1] = tid <= TKB_HARD ? ~0ull : 0ull;
    }
    __syncthreads();
    TK_PROF(0)
    // ---- B: classes + bitmaps, one wave per segment, lane = byte.
    // B1 (branch-free, all 17 segments of this wave in flight together): every lane finds the lead byte of
    // ITS char (0..3 bytes back), decodes the code point from the LDS copy of the text and issues the
    // stage-2 class load.  Continuation bytes therefore get their char's class without any cross-lane step.
    constexpr int NS = TK2_NSEG / 4;  // 17 segments per wave
    uint32_t creg[NS];
    {
        const uint32_t* dw = (const uint32_t*)raw;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const uint32_t pl = (uint32_t)(wid * NS + i) * 64u + lane;
            const uint32_t wi = pl >> 2, sft = pl & 3u;
            const uint32_t d0 = dw[wi ? wi - 1 : 0], d1 = dw[wi], d2 = dw[wi + 1];
            const uint32_t fwd = __builtin_amdgcn_alignbyte(d2, d1, sft);                        // bytes pl .. pl+3
            const uint32_t back = sft == 3u ? d1 : __builtin_amdgcn_alignbyte(d1, d0, sft + 1u);  // bytes pl-3 .. pl
            const uint32_t b = fwd & 0xFFu;
            uint32_t k = 0;
            if ((b & 0xC0u) == 0x80u) k = ((back >> 16) & 0xC0u) != 0x80u ? 1u : (((back >> 8) & 0xC0u) != 0x80u ? 2u : 3u);
            const uint64_t seven = ((uint64_t)fwd << 24) | (uint64_t)(back & 0xFFFFFFu);  // bytes pl-3 .. pl+3
            const uint32_t ch = (uint32_t)(seven >> (8u * (3u - k)));                      // the char's bytes, lead first
            const uint32_t l = ch & 0xFFu, c1b = (ch >> 8) & 0x3Fu, c2b = (ch >> 16) & 0x3Fu, c3b = (ch >> 24) & 0x3Fu;
            uint32_t cp = l;
            if (l >= 0xF0u) cp = ((l & 7u) << 18) | (c1b << 12) | (c2b << 6) | c3b;
            else if (l >= 0xE0u) cp = ((l & 15u) << 12) | (c1b << 6) | c2b;
            else if (l >= 0xC0u) cp = ((l & 31u) << 6) | c1b;
            if (cp > 0x10FFFFu) cp = 0xFFFFu;  // (U+FFFF is unassigned: class OTHER)
            creg[i] = T.uc_stage2[(uint32_t)st1[cp >> 8] * 256u + (cp & 255u)];
        }
    }
    // B2: flags, class bytes, bitmaps -- and, since a wave walks its 17 segments left to right, the certain
    // piece starts of the tile (class of the previous byte = lane - 1, carried across segments in a register)
    uint32_t carry = TK_C_END;
    if (wid > 0) {  // class of the byte just before this wave's first segment (same decode as B1, wave-uniform)
        const uint32_t* dw = (const uint32_t*)raw;
        const uint32_t pl = (uint32_t)(wid * NS) * 64u - 1u;
        const uint32_t wi = pl >> 2, sft = pl & 3u;
        const uint32_t d0 = dw[wi - 1], d1 = dw[wi], d2 = dw[wi + 1];
        const uint32_t fwd = __builtin_amdgcn_alignbyte(d2, d1, sft);
        const uint32_t back = sft == 3u ? d1 : __builtin_amdgcn_alignbyte(d1, d0, sft + 1u);
        const uint32_t b = fwd & 0xFFu;
        uint32_t k = 0;
        if ((b & 0xC0u) == 0x80u) k = ((back >> 16) & 0xC0u) != 0x80u ? 1u : (((back >> 8) & 0xC0u) != 0x80u ? 2u : 3u);
        const uint64_t seven = ((uint64_t)fwd << 24) | (uint64_t)(back & 0xFFFFFFu);
        const uint32_t ch = (uint32_t)(seven >> (8u * (3u - k)));
        const uint32_t l = ch & 0xFFu, c1b = (ch >> 8) & 0x3Fu, c2b = (ch >> 16) & 0x3Fu, c3b = (ch >> 24) & 0x3Fu;
        uint32_t cp = l;
        if (l >= 0xF0u) cp = ((l & 7u) << 18) | (c1b << 12) | (c2b << 6) | c3b;
        else if (l >= 0xE0u) cp = ((l & 15u) << 12) | (c1b << 6) | c2b;
        else if (l >= 0xC0u) cp = ((l & 31u) << 6) | c1b;
        if (cp > 0x10FFFFu) cp = 0xFFFFu;
        carry = T.uc_stage2[(uint32_t)st1[cp >> 8] * 256u + (cp & 255u)];
        const int64_t gpp = base + pl;
        if (gpp < 0 || (uint64_t)gpp >= n) carry = TK_C_END;
        else if ((ss && ((ssw[pl >> 5] >> (pl & 31u)) & 1u)) || (si && ((siw[pl >> 5] >> (pl & 31u)) & 1u))) carry = TK_C_SPEC;
    }
    uint32_t spill_mask = 0;  // bit i: this lane's certain start of segment i did not fit the list
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int g = wid * NS + i;
        const uint32_t pl = (uint32_t)g * 64u + lane;
        const int64_t gp = base + pl;
        const bool valid = gp >= 0 && (uint64_t)gp < n;
        const uint32_t b = raw[pl];
        bool cont = valid && (b & 0xC0u) == 0x80u;
        uint32_t c = creg[i];
        bool hard = false;
        if (!valid) {
            c = TK_C_END;
            hard = gp >= 0;  // past the end: look-ahead stops here
        } else {
            const bool spec_s = ss && ((ssw[pl >> 5] >> (pl & 31u)) & 1u), spec_i = si && ((siw[pl >> 5] >> (pl & 31u)) & 1u);
            if (spec_i) {
                cont = true;
                c = TK_C_SPEC;
            } else if (spec_s) {
                cont = false;
                c = TK_C_SPEC;
                hard = true;
            } else if (!cont) {
                hard = (brkw[pl >> 5] >> (pl & 31u)) & 1u;
            }
        }
        cls2[pl] = (uint8_t)(c | (cont ? 0x40u : 0u) | (hard ? 0x80u : 0u));
        {
            uint32_t prevc = __shfl_up(c, 1, 64);
            if (lane == 0) prevc = carry;
            carry = __shfl(c, 63, 64);
            const bool cert = valid && !cont && g >= 1 && g <= TK_TILE / 64 && (hard || ((certm[prevc] >> c) & 1u));
            const uint64_t certw = __ballot(cert);
            if (g >= 1 && g <= TK_TILE / 64) {
                if (lane == 0) {
                    bits[(g - 1) * 2] = (uint32_t)certw;
                    bits[(g - 1) * 2 + 1] = (uint32_t)(certw >> 32);
                }
                if (certw) {
                    uint32_t at = 0;
                    if (lane == 0) at = atomicAdd(&cn, (uint32_t)__popcll(certw));
                    at = __shfl(at, 0, 64) + (uint32_t)__popcll(certw & ((1ull << lane) - 1ull));
                    if (cert) {
                        if (at < TK2_CLIST) clist[at] = (uint16_t)pl;
                        else spill_mask |= 1u << i;
                    }
                }
            }
        }
        // only the bitmaps this pattern's alternatives use
        constexpr bool O2 = PAT == TK_PAT_O200K, R5 = PAT == TK_PAT_R50K;
        const uint64_t w_start = __ballot(!cont), w_hard = __ballot(hard);
        const uint64_t w_oth = __ballot((TK_M_OTHER >> c) & 1u), w_ws = __ballot((TK_M_WS >> c) & 1u), w_nu = __ballot(c == TK_C_NU);
        uint64_t w_L = 0, w_up = 0, w_low = 0, w_cas = 0, w_nl = 0, w_nlsl = 0;
        if constexpr (!O2) w_L = __ballot((TK_M_L >> c) & 1u);
        if constexpr (O2) {
            w_up = __ballot((TK_M_UPPERISH >> c) & 1u);
            w_low = __ballot((TK_M_LOWERISH >> c) & 1u);
            w_cas = __ballot(c == TK_C_LC || c == TK_C_MK);
            w_nlsl = __ballot(c == TK_C_NL || c == TK_C_SL);
        }
        if constexpr (!R5) w_nl = __ballot(c == TK_C_NL);
        if (lane == 0) {
            bm[TKB_START][g] = w_start;
            bm[TKB_HARD][g] = w_hard;
            bm[TKB_OTH][g] = w_oth;
            bm[TKB_WS][g] = w_ws;
            bm[TKB_NU][g] = w_nu;
            if constexpr (!O2) bm[TKB_L][g] = w_L;
            if constexpr (O2) {
                bm[TKB_UP][g] = w_up;
                bm[TKB_LOW][g] = w_low;
                bm[TKB_CAS][g] = w_cas;
                bm[TKB_NLSL][g] = w_nlsl;
            }
            if constexpr (!R5) bm[TKB_NL][g] = w_nl;
        }
    }
    __syncthreads();
    TK_PROF(1)
    constexpr int pat = PAT;
    TkWin2Acc acc{cls2, raw, base, &T, text, n, brk, ss, si};
    auto scan_from = [&](uint64_t p) {
        for (;;) {
            const int64_t r = (int64_t)p - base;
            uint32_t len = 0;
            if (r >= 0 && r + 64 <= TK2_WIN) {
                const uint32_t wi = (uint32_t)r >> 6, sh = (uint32_t)r & 63u;
                TkWin w;
#define TK_FUNNEL(kind) (sh ? ((bm[kind][wi] >> sh) | (bm[kind][wi + 1] << (64u - sh))) : bm[kind][wi])
                w.start = TK_FUNNEL(TKB_START);
                w.stop = TK_FUNNEL(TKB_HARD) & ~1ull;
                w.L = TK_FUNNEL(TKB_L);
                w.up = TK_FUNNEL(TKB_UP);
                w.low = TK_FUNNEL(TKB_LOW);
                w.cas = TK_FUNNEL(TKB_CAS);
                w.oth = TK_FUNNEL(TKB_OTH);
                w.ws = TK_FUNNEL(TKB_WS);
                w.nl = TK_FUNNEL(TKB_NL);
                w.nu = TK_FUNNEL(TKB_NU);
                w.nlsl = TK_FUNNEL(TKB_NLSL);
#undef TK_FUNNEL
                TkBmExt ext{bm, wi, sh, (uint32_t)(TK2_WIN - r)};
                len = tk_piece_len_bits(w, acc, ext, p, cls2[r] & 15u, pat);
            }
            uint64_t e = len ? p + len : tk_piece_end_slow(&acc, p, pat);
            if (e >= n) break;
            const int64_t re = (int64_t)e - base;
            uint32_t ce, pc;
            if (re < TK2_WIN) {
                ce = cls2[re];
                pc = cls2[re - 1] & 15u;
            } else {
                uint32_t both = tk_boundary_classes_slow(&acc, e);
                ce = both & 0xFFu;
                pc = both >> 8;
            }
            if (ce & 0x80u) break;
            if ((certm[pc] >> (ce & 15u)) & 1u) break;
            if (e < tile_start + TK_TILE)
                atomicOr(&bits[(uint32_t)(e - tile_start) >> 5], 1u << ((uint32_t)(e - tile_start) & 31));
            else
                atomicOr(&starts[e >> 5], 1u << (e & 31));
            p = e;
        }
    };
    // ---- D: one lane per certain start
    const uint32_t ncert = cn < TK2_CLIST ? cn : TK2_CLIST;
    for (uint32_t i = tid; i < ncert; i += 256) scan_from((uint64_t)(base + clist[i]));
    while (spill_mask) {
        int i = __ffs((int)spill_mask) - 1;
        spill_mask &= spill_mask - 1;
        scan_from((uint64_t)(base + (int64_t)((uint32_t)(wid * NS + i) * 64u + lane)));
    }
    __syncthreads();
    TK_PROF(3)
    // ---- E
    if (tid < TK_TILE / 32) {
        uint32_t v = bits[tid];
        if (v) atomicOr(&starts[tile_start / 32 + tid], v);
    }
    TK_PROF(4)
#undef TK_PROF
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

// single-workgroup exclusive scan of a (small) uint32 array in place; total -> total_out[0] (u64)
__global__ __launch_bounds__(1024) void tk_k_scan_small(uint32_t* __restrict__ a, uint64_t n, uint64_t* __restrict__ total_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint64_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int E = 8;  // consecutive elements per thread
    for (uint64_t base = 0; base < n; base += 1024 * E) {
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
        for (int w = 0; w < 16; ++w) {
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

__global__ __launch_bounds__(256) void tk_k_emit(const uint32_t* __restrict__ starts, uint64_t nwords, const uint32_t* __restrict__ blockpre,
                                                 uint32_t* __restrict__ pstart, uint64_t P, uint64_t n) {
    __shared__ uint32_t sh[8];
    uint64_t w = blockIdx.x * 256ull + threadIdx.x;
    uint32_t v = w < nwords ? starts[w] : 0;
    uint32_t tot;
    uint32_t ex = tk_block_exscan_256(__popc(v), &tot, sh);
    uint64_t o = (uint64_t)blockpre[blockIdx.x] + ex;
    while (v) {
        uint32_t b = __ffs(v) - 1;
        v &= v - 1;
        pstart[o++] = (uint32_t)(w * 32 + b);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pstart[P] = (uint32_t)n;
}

// ------------------------------------------------------------------------------------------
// whole-piece probe + per-lane merge of short pieces
// ------------------------------------------------------------------------------------------
#define TK_DUP_FLAG 0x80000000u  // cnt[p] = TK_DUP_FLAG | miss-table slot: resolved by tk_k_dup_fix

// Whole-piece probe (src/lib.rs:367).  Wave-autonomous: no workgroup barriers.  A lane handles TK_PPT
// pieces per iteration (coalesced piece offsets); hits write their token; misses try to claim a slot
// of the miss table -- a later identical piece finds the claim and only records the slot -- and the
// claimants are appended to the length-binned lists with one atomic per (wave, bin).
__global__ __launch_bounds__(256) void tk_k_lookup(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ pstart,
                                                   uint64_t P, const uint32_t* __restrict__ ss, uint32_t* __restrict__ tok1,
                                                   uint32_t* __restrict__ cnt, uint32_t* __restrict__ listM, TkBins bins,
                                                   uint32_t* __restrict__ listC, uint32_t* __restrict__ counters, TkMissTable mt, int dbg) {
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * (256 * TK_PPT); base < P; base += (uint64_t)gridDim.x * (256 * TK_PPT)) {
        uint32_t cat[TK_PPT];  // 0: done, 1 + bin: append to that bin, 1 + TK_NBIN: tree list
        uint32_t plen[TK_PPT];
#pragma unroll
        for (int k = 0; k < TK_PPT; ++k) {
            const uint64_t p = base + (uint64_t)k * 256 + tid;
            cat[k] = 0;
            plen[k] = 0;
            if (p >= P) continue;
            const uint32_t s = pstart[p], len = pstart[p + 1] - s;
            plen[k] = len;
            if (ss && tk_bit(ss, s)) {
                tok1[p] = tk_special_id(T, text, s, len);
                cnt[p] = 1;
                continue;
            }
            const uint64_t key = tk_key_of_text(text, s, len);
            uint32_t r = (dbg & 2) ? len : tk_probe_piece(T, key, len, [&](uint32_t off) { return tk_equal_bytes(text, s, T.tok_bytes, off, len); });
            if ((dbg & 8) && r == TK_RANK_MAX) r = 0;
            if (r != TK_RANK_MAX) {
                tok1[p] = r;
                cnt[p] = 1;
                continue;
            }
            if (len > TK_GLANE_MAX) {
                cat[k] = 1 + TK_NBIN;
                continue;
            }
            cat[k] = 1 + (uint32_t)tk_bin_of(len);
            if (mt.key && !(dbg & 256)) {
                unsigned long long kk = tk_mix64(key ^ ((uint64_t)len * 0xA24BAED4963EE407ull));
                if (dbg & 512) kk &= 0xFFFull;  // test hook: force collisions between different pieces
                if (kk == TK_EMPTY_KEY) kk = 0;
                uint32_t i = (uint32_t)(kk >> 7) & ((1u << TK_MT_BITS) - 1u);
                for (int t = 0; t < TK_MT_PROBES; ++t) {
                    unsigned long long cur = mt.key[i];
                    if (cur == TK_EMPTY_KEY) cur = atomicCAS(&mt.key[i], TK_EMPTY_KEY, kk);
                    if (cur == TK_EMPTY_KEY) {  // claimed: this piece is the one that gets merged
                        mt.rep[i] = (uint32_t)p;
                        break;
                    }
                    if (cur == kk) {  // an identical piece (to be verified) already claimed the slot
                        cnt[p] = TK_DUP_FLAG | i;
                        cat[k] = 0;
                        break;
                    }
                    i = (i + 1) & ((1u << TK_MT_BITS) - 1u);
                }
            }
        }
        // appends: rare once duplicates are filtered, so one atomic per (wave, bin) is enough
#pragma unroll
        for (int k = 0; k < TK_PPT; ++k) {
            if (!__ballot(cat[k] != 0)) continue;
            const uint64_t p = base + (uint64_t)k * 256 + tid;
            for (uint32_t b = 0; b < TK_NBIN; ++b) {
                const uint64_t m = __ballot(cat[k] == 1 + b);
                if (!m) continue;
                const int leader = __ffsll((unsigned long long)m) - 1;
                uint32_t at = 0;
                if (lane == leader) at = atomicAdd(&counters[TK_CNT_BIN0 + b], (uint32_t)__popcll(m));
                at = __shfl(at, leader, 64);
                if (cat[k] == 1 + b) listM[bins.off[b] + at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
            }
            if (cat[k] == 1 + TK_NBIN) {
                // scratch for the tree path: 4 uint32 per byte + the 64-ary min-tree levels
                uint32_t lv = 0, c = plen[k];
                do {
                    c = (c + 63) >> 6;
                    lv += c;
                } while (c > 64);
                const uint32_t gi = atomicAdd(&counters[TK_CNT_C], 1u);
                listC[3 * (uint64_t)gi] = (uint32_t)p;
                listC[3 * (uint64_t)gi + 1] = atomicAdd(&counters[TK_CNT_CBYTES], plen[k]);
                listC[3 * (uint64_t)gi + 2] = atomicAdd(&counters[TK_CNT_CLEVELS], lv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// One LANE per deferred piece of 2..64 bytes, lists binned by length so the 64 lanes of a wave run
// similar trip counts.  byte_pair_merge (src/lib.rs:140-196): ids and pair ranks of the piece's
// parts live in LDS (k-major, lane-minor: conflict-free), alive positions in a 128-bit register
// mask, so the only memory latency per merge is the pair of table probes.  What this buys over
// one-wave-per-piece is 64 independent probe chains per wavefront.
// ------------------------------------------------------------------------------------------
template <int NMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void tk_k_merge_llane(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ pstart,
                                                            const uint32_t* __restrict__ list, uint32_t count, uint32_t* __restrict__ tok1,
                                                            uint32_t* __restrict__ cnt, uint32_t* __restrict__ staging) {
    __shared__ uint32_t s_id[NMAX * THREADS];
    __shared__ uint32_t s_rk[NMAX * THREADS];
    uint32_t* id = s_id + threadIdx.x;
    uint32_t* rk = s_rk + threadIdx.x;
    for (uint32_t it = blockIdx.x * THREADS + threadIdx.x; it < count; it += gridDim.x * THREADS) {
        const uint32_t p = list[it];
        const uint32_t s = pstart[p], n = pstart[p + 1] - s;
        const uint32_t t = tk_lane_merge<THREADS>(T, text, s, n, id, rk, staging + s);
        cnt[p] = t;
        tok1[p] = t == 1 ? id[0] : p;  // multi-token results are fetched from staging[pstart[tok1[p]]]
    }
}

// ------------------------------------------------------------------------------------------
// G lanes per deferred piece (G = 8 / 16 / 32 / 64 for pieces of <= 128 / 256 / 512 / 1024 bytes).
// Lane g of a group owns part positions [16 g, 16 g + 16): ids and pair ranks in LDS, an alive
// bitmask and the cached minimum (rank << 32 | position) of its chunk in registers.  One merge =
// a log2(G)-step shuffle reduction of the cached minima (leftmost lowest rank, lib.rs:151,190),
// neighbour search through the alive masks, two pair probes (by two different lanes), and a
// re-scan of the <= 3 chunks that changed.  64/G pieces per wavefront, 8 KiB of LDS per wavefront.
// ------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void tk_k_merge_group(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ pstart,
                                                        const uint32_t* __restrict__ list, uint32_t count, uint32_t* __restrict__ tok1,
                                                        uint32_t* __restrict__ cnt, uint32_t* __restrict__ staging,
                                                        const uint32_t* __restrict__ count_ptr) {
    constexpr int C = 16, NMAX = G * C, PPW = 64 / G;
    if (count_ptr) count = *count_ptr;  // list length produced on the device (collision list)
    constexpr uint32_t NONE = 0xFFFFu;
    __shared__ __attribute__((aligned(16))) uint32_t s_id[4][1024];
    __shared__ __attribute__((aligned(16))) uint32_t s_rk[4][1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane & (G - 1), grp = lane / G, gbase = grp * G;
    uint32_t* id = s_id[wid] + grp * NMAX;
    uint32_t* rk = s_rk[wid] + grp * NMAX;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    auto local_min = [&]() -> uint64_t {
        uint64_t m = ~0ull;
        const uint4* q = (const uint4*)(rk + g * C);
#pragma unroll
        for (int v = 0; v < C / 4; ++v) {
            uint4 x = q[v];
            uint32_t k0 = g * C + v * 4;
            uint64_t a = ((uint64_t)x.x << 32) | k0, b = ((uint64_t)x.y << 32) | (k0 + 1), c = ((uint64_t)x.z << 32) | (k0 + 2),
                     d = ((uint64_t)x.w << 32) | (k0 + 3);
            a = a < b ? a : b;
            c = c < d ? c : d;
            a = a < c ? a : c;
            m = m < a ? m : a;
        }
        return m;
    };
    for (uint32_t e0 = wave * PPW; e0 < count; e0 += nwaves * PPW) {  // wave-uniform trip count
        const uint32_t e = e0 + grp;
        const bool valid = e < count;
        uint32_t p = 0, s = 0, n = 0;
        if (valid) {
            p = list[e];
            s = pstart[p];
            n = pstart[p + 1] - s;
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
        uint64_t lk = local_min();
        for (;;) {
            uint64_t m = lk;
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                uint64_t w = __shfl_xor(m, o, 64);
                m = w < m ? w : m;
            }
            const uint32_t best = (uint32_t)(m >> 32);
            const bool fin = best == TK_RANK_MAX;
            if (__all(fin)) break;
            // everything below is computed by every lane (shuffles must not sit under divergent control flow);
            // writes and probes are predicated on !fin
            const uint32_t bi = (uint32_t)m & (NMAX - 1), ob = bi / C, bl = bi % C;
            const uint64_t nbw = __ballot(mask != 0);
            const uint64_t nb = G == 64 ? nbw : ((nbw >> gbase) & ((1ull << (G & 63)) - 1ull));
            const uint32_t my_first = mask ? (uint32_t)(g * C + __ffs((int)mask) - 1) : NONE;
            const uint32_t my_last = mask ? (uint32_t)(g * C + 31 - __clz((int)mask)) : NONE;
            const uint32_t om = __shfl(mask, gbase + (int)ob, 64);
            // j: the part absorbed = next alive after bi
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
            // nn: next alive after j
            uint32_t nn;
            {
                const uint32_t ojm = __shfl(mask, gbase + (int)oj, 64);
                const uint32_t hi = ojm & ~((2u << jl) - 1u);
                const uint64_t la = nb & ~((2ull << oj) - 1ull);
                const int ln = la ? __ffsll((unsigned long long)la) - 1 : 0;
                const uint32_t fn = __shfl(my_first, gbase + ln, 64);
                nn = hi ? oj * C + (uint32_t)__ffs((int)hi) - 1u : (la ? fn : NONE);
            }
            // pp: previous alive before bi
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
            if (touched) lk = local_min();
        }
        // emit surviving parts, left to right
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
            if (g == 0) {
                cnt[p] = total;
                tok1[p] = total == 1 ? id[0] : p;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// one wavefront per piece longer than 64 bytes.  Parts form a doubly linked list in HBM scratch;
// a 64-ary tree of (rank << 32 | position) minima gives the lowest-rank, leftmost pair in one
// wave reduction over the root level; each merge touches <= 3 leaves and re-reduces their
// ancestors.  Same result as the heap formulation of src/lib.rs:47-138 (ordered by (rank, start)).
// ------------------------------------------------------------------------------------------
#define TK_MAX_LEVELS 6
__global__ __launch_bounds__(256) void tk_k_merge_long(TkTables T, const uint8_t* __restrict__ text, const uint32_t* __restrict__ pstart,
                                                       const uint32_t* __restrict__ listC, uint32_t nC, uint32_t* __restrict__ g_id,
                                                       uint32_t* __restrict__ g_rk, uint32_t* __restrict__ g_nx, uint32_t* __restrict__ g_pv,
                                                       uint64_t* __restrict__ g_lv, uint32_t* __restrict__ tok1, uint32_t* __restrict__ cnt,
                                                       uint32_t* __restrict__ staging) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (uint32_t w = wave; w < nC; w += nwaves) {
        const uint32_t p = listC[3 * (uint64_t)w];
        const uint32_t s = pstart[p], n = pstart[p + 1] - s;
        uint32_t* id = g_id + listC[3 * (uint64_t)w + 1];
        uint32_t* rk = g_rk + listC[3 * (uint64_t)w + 1];
        uint32_t* nx = g_nx + listC[3 * (uint64_t)w + 1];
        uint32_t* pv = g_pv + listC[3 * (uint64_t)w + 1];
        uint64_t* lv = g_lv + listC[3 * (uint64_t)w + 2];
        // level geometry: cntl[0] = n leaves (rk), cntl[l] = ceil(cntl[l-1]/64); the top level has <= 64 entries
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
            pv[k] = k - 1;  // k == 0 -> 0xFFFFFFFF (none)
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
        uint32_t ntok = n;
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
                rk[i] = newr;
            }
            if (lane == 1 && pp != 0xFFFFFFFFu) rk[pp] = newr;
            --ntok;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            // re-reduce the ancestors of leaves pp, i, j
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
        // emit the surviving parts in order
        if (ntok == 1) {
            if (lane == 0) {
                tok1[p] = id[0];
                cnt[p] = 1;
            }
        } else {
            if (lane == 0) {
                uint32_t t = 0;
                for (uint32_t k = 0; k < n; k = nx[k]) staging[s + t++] = id[k];
                cnt[p] = t;
                tok1[p] = p;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// exclusive scan of a large uint32 array (token counts): reduce / scan partials / downsweep
// ------------------------------------------------------------------------------------------
#define TK_SCAN_EPT 16  // elements per thread -> 4096 per workgroup
__global__ __launch_bounds__(256) void tk_k_scan_reduce(const uint32_t* __restrict__ a, uint64_t n, uint32_t* __restrict__ partial) {
    __shared__ uint32_t sh[8];
    uint64_t base = (uint64_t)blockIdx.x * (256 * TK_SCAN_EPT);
    uint32_t sum = 0;
    for (int k = 0; k < TK_SCAN_EPT; ++k) {
        uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) sum += a[i];
    }
    sum = tk_wave_sum_u32(sum);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[i] = exclusive prefix of a (out has n+1 entries; out[n] = total)
__global__ __launch_bounds__(256) void tk_k_scan_down(const uint32_t* __restrict__ a, uint64_t n, const uint32_t* __restrict__ partial_pre,
                                                      uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[8];
    uint64_t base = (uint64_t)blockIdx.x * (256 * TK_SCAN_EPT);
    uint32_t carry = partial_pre[blockIdx.x];
    for (int k = 0; k < TK_SCAN_EPT; ++k) {
        uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        uint32_t v = i < n ? a[i] : 0;
        uint32_t tot;
        uint32_t ex = tk_block_exscan_256(v, &tot, sh);
        if (i < n) out[i] = carry + ex;
        if (i + 1 == n) out[n] = carry + ex + v;
        carry += tot;
    }
}

// ------------------------------------------------------------------------------------------
// duplicates of a claimed missed piece: verify the bytes against the claimant and copy its result
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tk_k_dup_fix(const uint8_t* __restrict__ text, const uint32_t* __restrict__ pstart, uint64_t P,
                                                    TkMissTable mt, uint32_t* __restrict__ tok1, uint32_t* __restrict__ cnt,
                                                    uint32_t* __restrict__ coll_list, uint32_t* __restrict__ counters) {
    for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < P; p += (uint64_t)gridDim.x * 256) {
        const uint32_t c0 = cnt[p];
        if (!(c0 & TK_DUP_FLAG)) continue;
        const uint32_t rep = mt.rep[c0 & ~TK_DUP_FLAG];
        const uint32_t s = pstart[p], len = pstart[p + 1] - s, rs = pstart[rep], rlen = pstart[rep + 1] - rs;
        if (len == rlen && tk_equal_bytes(text, s, text, rs, len)) {
            const uint32_t c = cnt[rep];
            tok1[p] = c == 1 ? tok1[rep] : rep;
            cnt[p] = c;
        } else {  // different bytes behind the same 64-bit hash: encode this piece on its own
            cnt[p] = 0;
            coll_list[atomicAdd(&counters[TK_CNT_COLL], 1u)] = (uint32_t)p;
        }
    }
}

// ------------------------------------------------------------------------------------------
// final packing
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tk_k_gather(const uint32_t* __restrict__ pstart, uint64_t P, const uint32_t* __restrict__ cnt,
                                                   const uint32_t* __restrict__ tokbase, const uint32_t* __restrict__ tok1,
                                                   const uint32_t* __restrict__ staging, uint32_t* __restrict__ out) {
    for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < P; p += (uint64_t)gridDim.x * 256) {
        uint32_t c = cnt[p], b = tokbase[p];
        if (c == 1) {
            out[b] = tok1[p];
        } else {
            const uint32_t* src = staging + pstart[tok1[p]];  // own result, or the piece this one duplicates
            for (uint32_t t = 0; t < c; ++t) out[b + t] = src[t];
        }
    }
}

// tok_off[d] = number of tokens before document d  (= tokbase[#pieces that start before doc_off[d]])
__global__ __launch_bounds__(256) void tk_k_docoff(const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint64_t base, uint64_t n,
                                                   const uint32_t* __restrict__ starts, const uint32_t* __restrict__ blockpre,
                                                   const uint32_t* __restrict__ tokbase, uint64_t P, uint64_t tok_base_global,
                                                   uint64_t* __restrict__ tok_off) {
    for (uint64_t d = blockIdx.x * 256ull + threadIdx.x; d <= n_docs; d += (uint64_t)gridDim.x * 256) {
        uint64_t pos = doc_off[d] - base;
        uint64_t idx;
        if (pos >= n) {
            idx = P;
        } else {
            uint64_t w = pos >> 5, blk = w >> 8;
            idx = blockpre[blk];
            for (uint64_t k = blk << 8; k < w; ++k) idx += __popc(starts[k]);
            idx += __popc(starts[w] & ((1u << (pos & 31)) - 1u));
        }
        tok_off[d] = tok_base_global + tokbase[idx];
    }
}
