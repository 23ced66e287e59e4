// Micro-benchmark: issue rate of the integer vector / scalar instructions the encode kernels are made of (gfx950).
// Every kernel runs ITER iterations of 32 independent-ish instructions per wave; full occupancy (8 waves per SIMD).
// Output: wave-instructions per cycle per SIMD (VALU) or per CU (SALU), from wall time and the measured clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

// 8 accumulator registers a0..a7, operands b, c.  BODY is repeated 4 times per iteration (8 instr each).
#define KERNEL(NAME, ASM8)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                     \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 17, a7 = a0 ^ 19; \
        uint32_t b = seed | 1, c = seed >> 3;                                                      \
        for (int i = 0; i < ITER; ++i) {                                                            \
            asm volatile(ASM8 ASM8 ASM8 ASM8                                                        \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c)                                                           \
                         : "s10", "s11", "s12", "s13", "vcc", "scc");                                      \
        }                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }
#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")

#define OP_ADD(r) "v_add_u32 " r ", " r ", %8\n"
#define OP_AND(r) "v_and_b32 " r ", " r ", %8\n"
#define OP_XOR(r) "v_xor_b32 " r ", " r ", %8\n"
#define OP_SHL(r) "v_lshlrev_b32 " r ", 1, " r "\n"
#define OP_SHRV(r) "v_lshrrev_b32 " r ", %9, " r "\n"
#define OP_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define OP_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n"
#define OP_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define OP_XAD(r) "v_xad_u32 " r ", " r ", %8, %9\n"
#define OP_BFE(r) "v_bfe_u32 " r ", " r ", 3, 8\n"
#define OP_BFI(r) "v_bfi_b32 " r ", %8, " r ", %9\n"
#define OP_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define OP_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, %9\n"
#define OP_ALIGNBYTE(r) "v_alignbyte_b32 " r ", " r ", %8, %9\n"
#define OP_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define OP_MULHI(r) "v_mul_hi_u32 " r ", " r ", %8\n"
#define OP_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define OP_MUL24(r) "v_mul_u32_u24 " r ", " r ", %8\n"
#define OP_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define OP_CMP(r) "v_cmp_lt_u32 vcc, " r ", %8\n"
#define OP_CMPS(r) "v_cmp_lt_u32 s[10:11], " r ", %8\n"
#define OP_FFBL(r) "v_ffbl_b32 " r ", " r "\n"
#define OP_FFBH(r) "v_ffbh_u32 " r ", " r "\n"
#define OP_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8\n"
#define OP_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", " r ", %8\n"
#define OP_MOV(r) "v_mov_b32 " r ", %8\n"
#define OP_NOT(r) "v_not_b32 " r ", " r "\n"
#define OP_MIN(r) "v_min_u32 " r ", " r ", %8\n"
#define OP_MIN3(r) "v_min3_u32 " r ", " r ", %8, %9\n"
#define OP_SDWA(r) "v_lshlrev_b32_sdwa " r ", %9, " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define OP_DPP(r) "v_mov_b32_dpp " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_DPPWAVE(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ADDDPP(r) "v_add_u32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_READLANE(r) "v_readlane_b32 s10, " r ", 5\n"
#define OP_WRITELANE(r) "v_writelane_b32 " r ", s10, 5\n"
#define OP_READFIRST(r) "v_readfirstlane_b32 s10, " r "\n"
#define OP_FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define OP_ADDF(r) "v_add_f32 " r ", " r ", %8\n"
#define OP_SAND64(r) "s_and_b64 s[10:11], s[10:11], s[12:13]\n"
#define OP_SLSHL64(r) "s_lshl_b64 s[10:11], s[10:11], 1\n"
#define OP_SBCNT(r) "s_bcnt1_i32_b64 s10, s[12:13]\n"
#define OP_SFF1(r) "s_ff1_i32_b64 s10, s[12:13]\n"
#define OP_SADD(r) "s_add_u32 s10, s10, s12\n"
#define OP_SMOV(r) "s_mov_b32 s10, s12\n"
#define OP_BPERM(r) "ds_bpermute_b32 " r ", %8, " r "\ns_waitcnt lgkmcnt(0)\n"
#define OP_SWIZ(r) "ds_swizzle_b32 " r ", " r " offset:swizzle(SWAP,1)\ns_waitcnt lgkmcnt(0)\n"

KERNEL(k_add, R8(OP_ADD))
KERNEL(k_and, R8(OP_AND))
KERNEL(k_xor, R8(OP_XOR))
KERNEL(k_shl, R8(OP_SHL))
KERNEL(k_shrv, R8(OP_SHRV))
KERNEL(k_lshl_or, R8(OP_LSHLOR))
KERNEL(k_and_or, R8(OP_ANDOR))
KERNEL(k_add3, R8(OP_ADD3))
KERNEL(k_xad, R8(OP_XAD))
KERNEL(k_bfe, R8(OP_BFE))
KERNEL(k_bfi, R8(OP_BFI))
KERNEL(k_perm, R8(OP_PERM))
KERNEL(k_alignbit, R8(OP_ALIGNBIT))
KERNEL(k_alignbyte, R8(OP_ALIGNBYTE))
KERNEL(k_mul_lo, R8(OP_MULLO))
KERNEL(k_mul_hi, R8(OP_MULHI))
KERNEL(k_mad24, R8(OP_MAD24))
KERNEL(k_mul24, R8(OP_MUL24))
KERNEL(k_cndmask, R8(OP_CNDMASK))
KERNEL(k_cmp_vcc, R8(OP_CMP))
KERNEL(k_cmp_sgpr, R8(OP_CMPS))
KERNEL(k_ffbl, R8(OP_FFBL))
KERNEL(k_ffbh, R8(OP_FFBH))
KERNEL(k_bcnt, R8(OP_BCNT))
KERNEL(k_mbcnt, R8(OP_MBCNT))
KERNEL(k_mov, R8(OP_MOV))
KERNEL(k_not, R8(OP_NOT))
KERNEL(k_min, R8(OP_MIN))
KERNEL(k_min3, R8(OP_MIN3))
KERNEL(k_sdwa, R8(OP_SDWA))
KERNEL(k_dpp_row, R8(OP_DPP))
KERNEL(k_dpp_wave, R8(OP_DPPWAVE))
KERNEL(k_add_dpp, R8(OP_ADDDPP))
KERNEL(k_readlane, R8(OP_READLANE))
KERNEL(k_writelane, R8(OP_WRITELANE))
KERNEL(k_readfirstlane, R8(OP_READFIRST))
KERNEL(k_fma_f32, R8(OP_FMA))
KERNEL(k_add_f32, R8(OP_ADDF))
KERNEL(k_s_and64, R8(OP_SAND64))
KERNEL(k_s_lshl64, R8(OP_SLSHL64))
KERNEL(k_s_bcnt, R8(OP_SBCNT))
KERNEL(k_s_ff1, R8(OP_SFF1))
KERNEL(k_s_add, R8(OP_SADD))
KERNEL(k_s_mov, R8(OP_SMOV))
KERNEL(k_bpermute, R8(OP_BPERM))
KERNEL(k_swizzle, R8(OP_SWIZ))

// mixed: VALU + SALU interleaved (do they overlap?)
#define OP_MIXVS(r) "v_add_u32 " r ", " r ", %8\ns_add_u32 s10, s10, s12\n"
KERNEL(k_mix_valu_salu, R8(OP_MIXVS))
// 64-bit ops as the compiler emits them
__global__ __launch_bounds__(256) void k_shl64(uint32_t* out, uint32_t seed) {
    uint64_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = (uint64_t)(threadIdx.x + seed) * (0x9E3779B97F4A7C15ull + j);
    uint32_t s = seed & 31;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = (a[j] << s) | 1;
                asm volatile("" : "+v"(a[j]));
            }
    }
    uint64_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}
__global__ __launch_bounds__(256) void k_ctz64(uint32_t* out, uint32_t seed) {
    uint64_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = (uint64_t)(threadIdx.x + seed) * (0x9E3779B97F4A7C15ull + j);
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] += (uint64_t)__ffsll((unsigned long long)a[j]);
                asm volatile("" : "+v"(a[j]));
            }
    }
    uint64_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}
__global__ __launch_bounds__(256) void k_ballot(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x + seed) * (0x9E3779B9u + j);
    uint64_t acc = 0;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc += __ballot((a[j] >> (r + (i & 7))) & 1u);
            }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}
// LDS table gather: 256-entry u32 / u64 table, random byte index per lane
__global__ __launch_bounds__(256) void k_lds_gather32(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 0x01010101u + seed;
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x * 37 + seed) * (0x9E3779B9u + j);
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += tab[(a[j] >> 9) & 255u];
    }
    uint32_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_lds_gather64(uint32_t* out, uint32_t seed) {
    __shared__ uint64_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 0x0101010101010101ull + seed;
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x * 37 + seed) * (0x9E3779B9u + j);
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t t = tab[(a[j] >> 9) & 255u];
                a[j] += (uint32_t)t ^ (uint32_t)(t >> 32);
            }
    }
    uint32_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
// text-like gather: most lanes hit a handful of entries (ASCII letters)
__global__ __launch_bounds__(256) void k_lds_gather_text(uint32_t* out, uint32_t seed) {
    __shared__ uint64_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 0x0101010101010101ull + seed;
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x * 37 + seed) * (0x9E3779B9u + j);
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t t = tab[97u + ((a[j] >> 9) & 15u)];
                a[j] += (uint32_t)t ^ (uint32_t)(t >> 32);
            }
    }
    uint32_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Ent { const char* name; kern_t k; int instr_per_iter; };

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("device %s, %d CUs, clockRate %.3f GHz\n", p.name, cus, ghz);
    const int blocks = cus * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    uint32_t* out;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<Ent> es = {
#define E(n) {#n, n, 32}
        E(k_add), E(k_and), E(k_xor), E(k_shl), E(k_shrv), E(k_lshl_or), E(k_and_or), E(k_add3), E(k_xad), E(k_bfe), E(k_bfi), E(k_perm),
        E(k_alignbit), E(k_alignbyte), E(k_mul_lo), E(k_mul_hi), E(k_mad24), E(k_mul24), E(k_cndmask), E(k_cmp_vcc), E(k_cmp_sgpr), E(k_ffbl),
        E(k_ffbh), E(k_bcnt), E(k_mbcnt), E(k_mov), E(k_not), E(k_min), E(k_min3), E(k_sdwa), E(k_dpp_row), E(k_dpp_wave), E(k_add_dpp),
        E(k_readlane), E(k_writelane), E(k_readfirstlane), E(k_fma_f32), E(k_add_f32), E(k_s_and64), E(k_s_lshl64), E(k_s_bcnt), E(k_s_ff1),
        E(k_s_add), E(k_s_mov), E(k_bpermute), E(k_swizzle), {"k_mix_valu_salu(pairs)", k_mix_valu_salu, 32}, {"k_shl64(c++)", k_shl64, 32},
        {"k_ctz64(c++)", k_ctz64, 32}, {"k_ballot(c++)", k_ballot, 32}, {"k_lds_gather32", k_lds_gather32, 32}, {"k_lds_gather64", k_lds_gather64, 32},
        {"k_lds_gather_text64", k_lds_gather_text, 32},
    };
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    printf("%-28s %10s %14s %16s\n", "kernel", "ms", "cyc/op/SIMD", "ops/cyc/CU");
    for (auto& e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, a, b));
        // per SIMD: 8 waves * ITER * instr_per_iter wave-instructions
        const double ops_simd = 8.0 * ITER * e.instr_per_iter;
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("%-28s %10.3f %14.3f %16.3f\n", e.name, ms, cyc / ops_simd, ops_simd * 4 / cyc);
    }
    return 0;
}
