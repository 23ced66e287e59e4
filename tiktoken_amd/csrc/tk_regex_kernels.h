// Kernels of the generic pat_str engine (tk_regex.h: the matcher; tk_regex_split.h: what one lane does, shared with the CPU tests, which run it lane by lane).
//
//   tk_k_rx_speculate : one lane per segment (256 bytes or 1 KiB) of the chunk follows the chain of piece starts from the segment's first char as if it
//                       were a piece start (bitmap `spec`, exit position per segment);
//   tk_k_rx_resolve   : one lane per document walks the true chain, taking whole segments from `spec` wherever it lands on a speculative
//                       chain (exact: a match depends only on the text to its right) -> bitmap `gst` of true piece starts;
//   (gap chars -- positions at which the pattern matches nothing: find_iter skips them -- are pieces of their own, marked in `sgap` / `ggap`:
//    the front kernel gives them no token)
//   tk_k_rx_merge     : brk |= gst.  From here on every piece start is a "hard" start for the front kernel, which runs with a class table
//                       in which every char is a letter: its scanners then cut at hard starts and nowhere else, and everything behind the
//                       split -- whole-piece probe, de-duplication, merges, long pieces, token copy -- is the pipeline of the stock patterns.
//
// The program (<= 15 KiB) is copied to LDS by every workgroup; the property table (42 KiB) stays in global memory (L2).  Integer work,
// data-dependent branches, one lane per unit: this path is bound by divergence and latency, not by HBM -- it exists so that no pat_str
// is refused, the three stock families keep their hand-written scanners.
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
};

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

__global__ __launch_bounds__(256) void tk_k_rx_speculate(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                         const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si, uint32_t seg_shift,
                                                         uint32_t* __restrict__ spec, uint32_t* __restrict__ sgap, uint32_t* __restrict__ xexit) {
    __shared__ TkRxLds L;
    const TkRxProg P = tk_rx_stage_program(R, &L);
    const uint32_t nseg = (uint32_t)(((uint64_t)n + (1u << seg_shift) - 1u) >> seg_shift);
    const TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nseg; k += gridDim.x * blockDim.x) tk_rx_speculate_lane(P, t, k, seg_shift, spec, sgap, xexit);
}

__global__ __launch_bounds__(256) void tk_k_rx_resolve(TkRxDev R, const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ brk,
                                                       const uint32_t* __restrict__ ss, const uint32_t* __restrict__ si,
                                                       const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint64_t base, uint32_t seg_shift,
                                                       const uint32_t* __restrict__ spec, const uint32_t* __restrict__ sgap, const uint32_t* __restrict__ xexit,
                                                       uint32_t* __restrict__ gst, uint32_t* __restrict__ ggap, uint32_t* __restrict__ counters) {
    __shared__ TkRxLds L;
    const TkRxProg P = tk_rx_stage_program(R, &L);
    const TkRxText t{text, n, brk, ss, si, 0xFFFFFFFFu, false};
    for (uint64_t d = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; d < n_docs; d += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = doc_off[d] - base, e = doc_off[d + 1] - base;
        if (b >= e || e > n) continue;
        uint32_t err_pos = 0;
        const uint32_t err = tk_rx_resolve_lane(P, t, (uint32_t)b, (uint32_t)e, seg_shift, spec, sgap, xexit,
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

__global__ void tk_k_rx_merge(uint32_t* __restrict__ brk, const uint32_t* __restrict__ gst, uint64_t nwords) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) brk[i] |= gst[i];
}
