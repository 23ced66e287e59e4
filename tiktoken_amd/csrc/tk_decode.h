// Decode on the device: token ids -> bytes (reference CoreBPE::decode_bytes, src/lib.rs:345-358: concatenate decoder[token] for every
// token; Encoding.decode_batch / decode_bytes_batch, tiktoken/core.py:331-350, map it over documents).  A gather plus a prefix sum:
//   tk_k_dec_len     length of every token from the id -> {offset, length} table (an id without entry raises the reference's KeyError)
//                    and the byte count of every workgroup of 2048 tokens
//   tk_k_dec_scan64  exclusive prefix sum of those counts (64-bit: outputs may exceed 4 GiB)
//   tk_k_dec_copy    workgroup base + local scan = byte offset of every token; one lane per token copies its bytes to their place
//   tk_k_dec_docoff  byte offset of every document of a packed batch
// Included by tk_api.hip only.
#pragma once
#include "tk_kernels.h"
#include "tk_device.h"

#define TK_DEC_SPEC 0x80000000u  // entry.x bit: the bytes live in the special-token blob
#define TK_DEC_BLOCK 2048        // tokens per workgroup of the scan passes (256 threads x 8)

__global__ __launch_bounds__(256) void tk_k_dec_len(const uint32_t* __restrict__ tokens, uint64_t n, const uint2* __restrict__ dec,
                                                    uint32_t n_ids, uint32_t* __restrict__ lens, unsigned long long* __restrict__ bsum,
                                                    unsigned long long* __restrict__ bad /* position of the first token without entry; ~0 = none */,
                                                    uint64_t pos_base /* position of tokens[0] in the caller's batch (a batch is decoded in ranges) */) {
    __shared__ uint32_t sh[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * TK_DEC_BLOCK + (uint64_t)threadIdx.x * 8;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t i = i0 + j;
        uint32_t len = 0;
        if (i < n) {
            const uint32_t t = tokens[i];
            len = t < n_ids ? dec[t].y : 0u;
            if (len == 0u) atomicMin(bad, (unsigned long long)(pos_base + i));  // (every real token has at least one byte)
            lens[i] = len;
        }
        sum += len;
    }
    sum = tk_wave_sum_u32(sum);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = (unsigned long long)sh[0] + sh[1] + sh[2] + sh[3];
}

// single-workgroup exclusive scan of the per-workgroup sums (64-bit); total -> total_out[0]
__global__ __launch_bounds__(1024) void tk_k_dec_scan64(unsigned long long* __restrict__ a, uint64_t nb, unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (uint64_t base = 0; base < nb; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const unsigned long long v = i < nb ? a[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long w = __shfl_up(inc, o, 64);
            if (lane >= o) inc += w;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        unsigned long long wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wid) wbase += wsum[w];
            tot += wsum[w];
        }
        const unsigned long long carry = carry_sh;
        if (i < nb) a[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_sh = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total_out[0] = carry_sh;
}

// byte offset of every token: workgroup base + local exclusive scan; then the copy.  A lane owns eight consecutive tokens, so its bytes
// are one contiguous stretch of the output: they stream through a 64-bit register aligned with the destination -- one aligned 8-byte
// store per eight bytes (the first and the last word of the stretch, shared with the neighbours, go out byte by byte), sources read as
// aligned words and shifted (tk_load8: the blobs are readable 16 bytes past their ends).
__global__ __launch_bounds__(256) void tk_k_dec_copy(const uint32_t* __restrict__ tokens, uint64_t n, const uint2* __restrict__ dec,
                                                     const uint32_t* __restrict__ lens, const unsigned long long* __restrict__ bbase,
                                                     const uint8_t* __restrict__ tok_bytes, const uint8_t* __restrict__ spec_bytes,
                                                     uint8_t* __restrict__ out, unsigned long long* __restrict__ tok_byte_off /* may be null */,
                                                     unsigned long long off_base /* bytes of the batch before this range (added to tok_byte_off only) */) {
    __shared__ uint32_t sh[8];
    const uint64_t i0 = (uint64_t)blockIdx.x * TK_DEC_BLOCK + (uint64_t)threadIdx.x * 8;
    uint32_t len[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        len[j] = i0 + j < n ? lens[i0 + j] : 0u;
        mine += len[j];
    }
    uint32_t tot;
    const uint32_t ex = tk_block_exscan_256(mine, &tot, sh);
    unsigned long long at = bbase[blockIdx.x] + ex;
    // the stream: `acc` holds the bytes of the aligned word at `word` from byte `fill` on (bytes below `first_lo` of the first word are
    // a neighbour's)
    unsigned long long word = at & ~7ull;
    const uint32_t first_lo = (uint32_t)(at & 7ull);
    uint32_t fill = first_lo;
    uint64_t acc = 0;
    bool first = true;
    auto flush = [&](uint32_t upto /* bytes of the word that are valid: fill */) {
        if (!first && upto == 8u) {
            *(uint64_t*)(out + word) = acc;
        } else {
            for (uint32_t b = first ? first_lo : 0u; b < upto; ++b) out[word + b] = (uint8_t)(acc >> (8u * b));
        }
        first = false;
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t i = i0 + j;
        if (i < n) {
            if (tok_byte_off) tok_byte_off[i] = off_base + at;
            const uint2 e = dec[tokens[i]];
            const uint8_t* src = ((e.x & TK_DEC_SPEC) ? spec_bytes : tok_bytes);
            const uint64_t so = e.x & ~TK_DEC_SPEC;
            for (uint32_t b = 0; b < len[j]; b += 8u) {
                const uint32_t nb = len[j] - b < 8u ? len[j] - b : 8u;
                const uint64_t w = tk_mask_low_bytes(tk_load8(src, so + b), nb);
                acc |= w << (8u * fill);
                if (fill + nb >= 8u) {
                    flush(8u);
                    acc = fill ? (w >> (8u * (8u - fill))) : 0ull;
                    word += 8ull;
                    fill = fill + nb - 8u;
                } else {
                    fill += nb;
                }
            }
            at += len[j];
        }
    }
    if (fill > (first ? first_lo : 0u)) flush(fill);
}

// byte_off[d] = byte offset of the first token of document d (tok_off: n_docs + 1 token offsets, non-decreasing)
__global__ __launch_bounds__(256) void tk_k_dec_docoff(const uint64_t* __restrict__ tok_off, uint64_t n_docs, uint64_t n,
                                                       const unsigned long long* __restrict__ tok_byte_off,
                                                       const unsigned long long* __restrict__ total, uint64_t* __restrict__ byte_off) {
    for (uint64_t d = blockIdx.x * 256ull + threadIdx.x; d <= n_docs; d += (uint64_t)gridDim.x * 256) {
        const uint64_t t = tok_off[d];
        byte_off[d] = t < n ? tok_byte_off[t] : total[0];
    }
}
