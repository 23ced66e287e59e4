// Kernels of the generic pat_str engine (tk_regex.h: the matcher; tk_regex_split.h: what one lane does, shared with the CPU tests, which run it lane by lane).
//
//   tk_k_rx_speculate : one lane per segment (256 bytes or 1 KiB) of the chunk follows the chain of piece starts from the segment's first char as if it
//                       were a piece start (bitmap `spec`, exit position per segment);
//   tk_k_rx_link      : one lane per segment walks from where the segment before it was left (as guessed) until it meets its own segment's
//                       chain: with these links a segment is taken whole wherever the guesses were right;
//   tk_k_rx_resolve_wave : one wavefront per document walks the true chain, 64 segments per step wherever the plans of its lanes hold, one
//                       match where they do not (exact: a match depends only on the text around it) -> bitmap `gst` of true piece starts
//                       (tk_k_rx_resolve: the same by one lane per document);
//   (gap chars -- positions at which the pattern matches nothing: find_iter skips them -- are pieces of their own, marked in `sgap` / `ggap`:
//    the front kernel gives them no token)
//   tk_k_rx_merge     : brk |= gst.  From here on every piece start is a "hard" start for the front kernel, which runs with a class table
//                       in which every char is a letter: its scanners then cut at hard starts and nowhere else, and everything behind the
//                       split -- whole-piece probe, de-duplication, merges, long pieces, token copy -- is the pipeline of the stock patterns.
//
// Two forms of every kernel (template parameter FORM).  Where the pattern has a DFA (tk_regex_dfa.inc: the stock patterns, Qwen / Llama /
// DeepSeek / Kimi-style ones -- anything without look-behind, \b, look-ahead of several chars or atomic groups around groups) the matcher
// is one table look-up per char: the transition table (states x classes x 2 bytes: 2 KiB for o200k's pat_str) and the ASCII class table sit
// in LDS, the classes of non-ASCII chars come from a two-stage table in global memory (L2); every lane of a wavefront runs the same loop,
// and the speculative pass is ONE loop over the chars of a lane's segment (tk_rx_speculate_lane_flat).  Otherwise the backtracking program
// (<= 19 KiB) is copied to LDS by every workgroup and interpreted, the property table (42 KiB) stays in global memory: integer work,
// data-dependent branches, every lane in a different instruction -- bound by divergence, it exists so that no pat_str is refused.
#pragma once
#include "tk_regex_host.h"
#include "tk_regex_split.h"

struct TkRxDev {  // the compiled program in device memory
    const TkRxIns* ins;
    const TkRxSet* sets;
    const uint32_t* ranges;
    const uint8_t* stage1;
    const uint8_t* stage2;
    uint32_t n_ins, n_sets, n_ranges;
    const uint32_t* first;
    uint32_t n_first;
    // the DFA (null: none): the transition table padded to whole 32-bit words, the ASCII classes (128 bytes), the two-stage class table
    const uint16_t* dfa_trans;
    const uint8_t* dfa_ascii;
    const uint16_t* dfa_s1;
    const uint8_t* dfa_s2;
    uint32_t dfa_ncls, dfa_nstates, dfa_flags;
};
// (FLAT: the speculative pass as one loop, the other kernels as DFA; _PREV: the table of a pattern that looks behind -- its matcher reads the
// char in front of a match, an instantiation of its own so that the others do not carry that code)
enum { TK_RX_FORM_PROGRAM = 0, TK_RX_FORM_DFA = 1, TK_RX_FORM_DFA_FLAT = 2, TK_RX_FORM_DFA_PREV = 3, TK_RX_FORM_DFA_FLAT_PREV = 4 };
constexpr int tk_rx_matcher_of(int form) { return form == TK_RX_FORM_PROGRAM ? TK_RX_M_PROGRAM : (form >= TK_RX_FORM_DFA_PREV ? TK_RX_M_DFA_PREV : TK_RX_M_DFA); }
// bytes of dynamic LDS the DFA forms need
static inline uint32_t tk_rx_dfa_lds_bytes(const TkRxDev& R) { return ((R.dfa_nstates * R.dfa_ncls + 1u) / 2u) * 4u + 384u + 0x1100u * 2u; }

struct TkRxLds {
    TkRxIns ins[TK_RX_MAX_INS];
    TkRxSet sets[TK_RX_MAX_SETS];
    uint32_t ranges[2 * TK_RX_MAX_RANGES];
    uint32_t first[8 * TK_RX_MAX_FIRST];
};

__device__ __forceinline__ TkRxProg tk_rx_stage_program(const TkRxDev& R, TkRxLds* L) {
    const uint32_t* si = (const uint32_t*)R.ins;
    uint32_t* di = (uint32_t*)L->ins;
    for (uint32_t i = threadIdx.x; i < R.n_ins * 4u; i += blockDim.x) di[i] = si[i];
    const uint32_t* ss = (const uint32_t*)R.sets;
    uint32_t* ds = (uint32_t*)L->sets;
    for (uint32_t i = threadIdx.x; i < R.n_sets * 8u; i += blockDim.x) ds[i] = ss[i];
    for (uint32_t i = threadIdx.x; i < R.n_ranges * 2u; i += blockDim.x) L->ranges[i] = R.ranges[i];
    for (uint32_t i = threadIdx.x; i < R.n_first * 8u; i += blockDim.x) L->first[i] = R.first[i];
    __syncthreads();
    return TkRxProg{L->ins, L->sets, L->ranges, R.stage1, R.stage2, R.n_ins, R.n_sets, R.n_ranges, L->first, R.n_first};
}

// the DFA's tables in (dynamic) LDS
__device__ __forceinline__ TkRxProg tk_rx_stage_dfa(const TkRxDev& R, uint32_t* lds) {
    // (the ASCII classes, then the first stage of the class table, then the transitions: all at constant offsets from the start of LDS,
    // so that every look-up is a ds_read)
    const uint32_t nt = (R.dfa_nstates * R.dfa_ncls + 1u) / 2u;
    const uint32_t* st = (const uint32_t*)R.dfa_trans;
    const uint32_t* sa = (const uint32_t*)R.dfa_ascii;
    const uint32_t* s1 = (const uint32_t*)R.dfa_s1;
    for (uint32_t i = threadIdx.x; i < 96u; i += blockDim.x) lds[i] = sa[i];  // (128 ASCII classes + 256 groups of classes)
    for (uint32_t i = threadIdx.x; i < 0x880u; i += blockDim.x) lds[96u + i] = s1[i];
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) lds[96u + 0x880u + i] = st[i];
    __syncthreads();
    TkRxProg P{};
    P.dfa_ascii = (const uint8_t*)lds;
    P.dfa_s1 = (const uint16_t*)(lds + 96);
    P.dfa_trans = (const uint16_t*)(lds + 96 + 0x880);
    P.dfa_flags = R.dfa_flags;
    P.dfa_s2 = R.dfa_s2;  // (the second stage: global memory, L2)
    P.dfa_ncls = R.dfa_ncls;
    return P;
}
extern __shared__ uint32_t tk_rx_dyn_lds[];
#define TK_RX_STAGE(P, R)                                  \
    TkRxProg P;                                            \
    if constexpr (FORM != TK_RX_FORM_PROGRAM) {            \
        P = tk_rx_stage_dfa(R, tk_rx_dyn_lds);             \
    } else {                                               \
        __shared__ TkRxLds L;                              \
        P = tk_rx_stage_program(R, &L);                    \
    }

template <int FORM>
__global__ __launch_bounds__(256) void tk_k_rx_speculate(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                         const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si, uint32_t seg_shift, uint32_t ahead,
                                                         uint32_t* __restrict__ spec, uint32_t* __restrict__ sgap, uint32_t* __restrict__ xexit) {
    TK_RX_STAGE(P, R)
    const uint32_t nseg = (uint32_t)(((uint64_t)n + (1u << seg_shift) - 1u) >> seg_shift);
    TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
    t.ahead = ahead;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nseg; k += gridDim.x * blockDim.x) {
        if constexpr (FORM == TK_RX_FORM_DFA_FLAT) tk_rx_speculate_lane_flat<false>(P, t, k, seg_shift, spec, sgap, xexit);
        else if constexpr (FORM == TK_RX_FORM_DFA_FLAT_PREV) tk_rx_speculate_lane_flat<true>(P, t, k, seg_shift, spec, sgap, xexit);
        else tk_rx_speculate_lane<tk_rx_matcher_of(FORM)>(P, t, k, seg_shift, spec, sgap, xexit);
    }
}

// The speculative pass over STAGED text (tk_regex_split.h, "The speculative pass over STAGED text"): a workgroup turns the 32 KiB of its 256
// segments into one code per byte in LDS -- 16 bytes per lane and step: 4 KiB per load instruction, coalesced, every char decoded and
// classified once -- and its lanes then walk codes: two LDS reads per char (the code, the transition), the bits of the segment's words in
// registers, one store at the end; nothing in the loop waits for global memory.  LDS: the ASCII classes and the transitions as in the
// other kernels (the first stage of the class table stays in global memory: only phase 1 reads it, for chars beyond ASCII), then the codes.
// For the pattern's DFA without look-behind, 128-byte segments and at most TK_RX_CODE_MAX_CLS + 1 classes (tk_api.hip, rx_split).
static inline uint32_t tk_rx_staged_lds_bytes(const TkRxDev& R) { return 384u + ((R.dfa_nstates * R.dfa_ncls + 1u) / 2u) * 4u + TK_RX_STAGE_BYTES; }
static inline bool tk_rx_staged_fits(const TkRxDev& R) {
    return R.dfa_trans && !(R.dfa_flags & 1u) && R.dfa_ncls <= TK_RX_CODE_MAX_CLS + 1u && tk_rx_staged_lds_bytes(R) <= 56u * 1024u && R.dfa_nstates * R.dfa_ncls < 32768u;
}

__global__ __launch_bounds__(TK_RX_STAGE_SEGS) void tk_k_rx_speculate_staged(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                                const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si, uint32_t ahead,
                                                                uint32_t* __restrict__ spec, uint32_t* __restrict__ sgap, uint32_t* __restrict__ xexit) {
    uint32_t* lds = tk_rx_dyn_lds;
    const uint32_t nt = (R.dfa_nstates * R.dfa_ncls + 1u) / 2u;
    {
        const uint32_t* sa = (const uint32_t*)R.dfa_ascii;
        for (uint32_t i = threadIdx.x; i < 96u; i += blockDim.x) lds[i] = sa[i];  // (128 ASCII classes + 256 groups of classes)
        uint16_t* tp = (uint16_t*)(lds + 96);  // the transitions with row offsets for state numbers (tk_rx_trans_premultiplied)
        for (uint32_t i = threadIdx.x; i < 2u * nt; i += blockDim.x) tp[i] = tk_rx_trans_premultiplied(R.dfa_trans[i], R.dfa_ncls);
    }
    TkRxProg P{};  // (for the codes -- the class tables -- and for the one-loop lane: its transitions, like the two stages of the class table, in global memory)
    P.dfa_ascii = (const uint8_t*)lds;
    P.dfa_trans = R.dfa_trans;
    P.dfa_s1 = R.dfa_s1;
    P.dfa_s2 = R.dfa_s2;
    P.dfa_flags = R.dfa_flags;
    P.dfa_ncls = R.dfa_ncls;
    const uint16_t* trans_pm = (const uint16_t*)(lds + 96);
    uint8_t* codes = (uint8_t*)(lds + 96u + nt);
    const uint32_t nseg = (uint32_t)(((uint64_t)n + (1u << TK_RX_SEG_SHIFT_SMALL) - 1u) >> TK_RX_SEG_SHIFT_SMALL);
    TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
    t.ahead = ahead;
    for (uint32_t s0 = blockIdx.x * TK_RX_STAGE_SEGS; s0 < nseg; s0 += gridDim.x * TK_RX_STAGE_SEGS) {
        const uint32_t r0 = s0 << TK_RX_SEG_SHIFT_SMALL;
        __syncthreads();  // (the tables are in place; the lanes of the stretch before are done with its codes)
        for (uint32_t b = threadIdx.x * 16u; b < TK_RX_STAGE_BYTES; b += TK_RX_STAGE_SEGS * 16u) {
            uint32_t o[4] = {0u, 0u, 0u, 0u};
            if ((uint64_t)r0 + b < n) tk_rx_codes16(P, t, r0 + b, o);
            tk_rx_codes_store(codes, b, o);
        }
        __syncthreads();
        const uint32_t k = s0 + threadIdx.x;
        if (k < nseg) {
            const TkRxCodes C{codes, r0, TK_RX_STAGE_BYTES};
            uint32_t sb[4], gb[4], x;
            if (tk_rx_speculate_lane_codes(trans_pm, R.dfa_ncls, C, n, ahead, k, sb, gb, &x)) {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {  // (the bitmaps start out zero and these words are the lane's: plain stores; a bit is a position of the text)
                    if (sb[i]) spec[4u * (size_t)k + i] = sb[i];
                    if (gb[i]) sgap[4u * (size_t)k + i] = gb[i];
                }
                xexit[k] = x;
            } else {
                tk_rx_speculate_lane_flat<false>(P, t, k, TK_RX_SEG_SHIFT_SMALL, spec, sgap, xexit);
            }
        }
    }
}

template <int FORM>
__global__ __launch_bounds__(256) void tk_k_rx_link(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                    const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si, uint32_t seg_shift, uint32_t ahead,
                                                    const uint32_t* __restrict__ spec, const uint32_t* __restrict__ xexit, uint32_t* __restrict__ lnk,
                                                    uint32_t* __restrict__ lgap, uint32_t* __restrict__ lmerge, uint32_t* __restrict__ lexit) {
    TK_RX_STAGE(P, R)
    const uint32_t nseg = (uint32_t)(((uint64_t)n + (1u << seg_shift) - 1u) >> seg_shift);
    TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
    t.ahead = ahead;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nseg; k += gridDim.x * blockDim.x)
        tk_rx_link_lane<tk_rx_matcher_of(FORM)>(P, t, k, seg_shift, spec, xexit, lnk, lgap, lmerge, lexit);
}

// one lane per document (debug bit 0x40000; the CPU tests run this form lane by lane)
template <int FORM>
__global__ __launch_bounds__(256) void tk_k_rx_resolve(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                       const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si,
                                                       const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint64_t base, TkRxMaps M,
                                                       uint32_t* __restrict__ gst, uint32_t* __restrict__ ggap, uint32_t* __restrict__ counters) {
    TK_RX_STAGE(P, R)
    const TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
    for (uint64_t d = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; d < n_docs; d += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = doc_off[d] - base, e = doc_off[d + 1] - base;
        if (b >= e || e > n) continue;
        uint32_t err_pos = 0;
        const uint32_t err = tk_rx_resolve_lane<tk_rx_matcher_of(FORM)>(P, t, M, (uint32_t)b, (uint32_t)e,
                                                [&](uint32_t w, uint32_t bits, uint32_t gaps) {
                                                    if (bits) atomicOr(&gst[w], bits);
                                                    if (gaps) atomicOr(&ggap[w], gaps);
                                                },
                                                &err_pos);
        if (err) {
            atomicOr(&counters[TK_CNT_ERR], err);
            atomicMax(&counters[TK_CNT_RXPOS], ~err_pos);  // (the counters start at zero: the smallest position wins)
        }
    }
}

// One WAVEFRONT per document (tk_rx_resolve_group_host is the same, lane after lane): lane j plans segment k0 + j as if the chain entered
// it where the guess for the segment before it ended; the longest prefix of lanes whose plans hold -- every exit known and equal to the next
// lane's entry -- is taken at once, 64 segments per step.  Where lane 0 has no plan the wavefront takes one step of the serial form (all
// lanes run it on the same arguments: uniform control flow, lane 0 writes).  A single document of many megabytes is resolved by its
// wavefront at the rate of its memory operations, not of one lane's matcher.
template <int FORM>
__global__ __launch_bounds__(256) void tk_k_rx_resolve_wave(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                            const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si,
                                                            const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint64_t base, TkRxMaps M,
                                                            uint32_t* __restrict__ gst, uint32_t* __restrict__ ggap, uint32_t* __restrict__ counters) {
    TK_RX_STAGE(P, R)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    auto orb_all = [&](uint32_t w, uint32_t bits, uint32_t gaps) {
        if (bits) atomicOr(&gst[w], bits);
        if (gaps) atomicOr(&ggap[w], gaps);
    };
    auto orb_lane0 = [&](uint32_t w, uint32_t bits, uint32_t gaps) {
        if (lane == 0u) orb_all(w, bits, gaps);
    };
    for (uint64_t d = wave; d < n_docs; d += nwaves) {
        const uint64_t b64 = doc_off[d] - base, e64 = doc_off[d + 1] - base;
        if (b64 >= e64 || e64 > n) continue;
        const uint32_t e = (uint32_t)e64;
        TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
        uint32_t p = (uint32_t)b64, err = 0;
        while (p < e) {
            const uint32_t k = (p >> M.seg_shift) + lane;
            uint32_t entry = p;
            if (lane) entry = ((uint64_t)k << M.seg_shift) < n ? M.xexit[k - 1u] : TK_RX_UNKNOWN;
            TkRxPlan plan{false, false, 0, 0, TK_RX_UNKNOWN};
            if (entry != TK_RX_UNKNOWN) plan = tk_rx_plan(M, n, k, entry, e);
            // lanes 0 .. len - 1: every plan holds, every exit but the last is known and is the next lane's entry
            const uint32_t prev_exit = (uint32_t)__shfl_up((int)plan.exit, 1, 64);
            const bool chained = plan.ok && (lane == 0u || (prev_exit != TK_RX_UNKNOWN && prev_exit == entry));
            const uint64_t bad = ~__ballot(chained);
            uint32_t len = bad ? (uint32_t)__ffsll((unsigned long long)bad) - 1u : 64u;
            if (len && (uint32_t)__shfl((int)plan.exit, (int)len - 1, 64) == TK_RX_UNKNOWN) --len;  // (that segment's guess broke off: the serial step's)
            if (len == 0u) {
                if constexpr (FORM != TK_RX_FORM_PROGRAM) {
                    // the wavefront matches the piece together: the table walk, and for a long run a KiB per step (tk_rx_match_dfa_coop)
                    auto coop = [&](uint32_t S, uint32_t start, uint32_t pos, uint32_t base, uint32_t* pbad, uint32_t* m1, uint32_t* pnext) {
                        uint32_t bad, mat, endp;
                        tk_rx_run_lane(P, t, S, start, pos, base + 16u * lane, &bad, &mat, &endp);
                        *pnext = (uint32_t)__shfl((int)endp, 63, 64);
                        uint32_t pb = bad;
                        for (int o = 32; o > 0; o >>= 1) {
                            const uint32_t v = (uint32_t)__shfl_xor((int)pb, o, 64);
                            pb = v < pb ? v : pb;
                        }
                        uint32_t m = (mat != TK_RX_NONE && mat < pb) ? mat + 1u : 0u;
                        for (int o = 32; o > 0; o >>= 1) {
                            const uint32_t v = (uint32_t)__shfl_xor((int)m, o, 64);
                            m = v > m ? v : m;
                        }
                        *pbad = pb;
                        *m1 = m;
                    };
                    p = tk_rx_resolve_step_with(P, t, M, p, e, orb_lane0, &err, [&](uint32_t at) { return tk_rx_match_dfa_coop<tk_rx_matcher_of(FORM) == TK_RX_M_DFA_PREV>(P, t, at, coop); });
                } else {
                    p = tk_rx_resolve_step<false>(P, t, M, p, e, orb_lane0, &err);
                }
                if (err) break;
                continue;
            }
            if (lane < len) (void)tk_rx_emit(M, plan, entry, orb_all);
            p = (uint32_t)__shfl((int)plan.exit, (int)len - 1, 64);
        }
        if (err && lane == 0u) {
            atomicOr(&counters[TK_CNT_ERR], err);
            atomicMax(&counters[TK_CNT_RXPOS], ~p);  // (the counters start at zero: the smallest position wins)
        }
    }
}

__global__ void tk_k_rx_merge(uint32_t* __restrict__ brk, const uint32_t* __restrict__ gst, uint64_t nwords) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) brk[i] |= gst[i];
}
