// Round 6: what the link between host and device gives on the box the bench runs on -- the ceiling of the host-buffer entry
// (tk_encode_batch, "T2": pageable host text in, token ids in host memory out).  Page-locked H2D, D2H, both at once (two streams),
// in blocks of 128 MiB over 1 GiB each way; beside them what stands between a caller's pageable buffer and the link: the parallel
// memcpy into a page-locked staging buffer (1 .. 32 threads), hipMemcpy straight from pageable memory, and hipHostRegister of the
// caller's buffer (the time to pin 1 GiB, and the DMA rate from it).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/pcie_rates.hip -pthread -o tools/ubench/pcie_rates && tools/ubench/pcie_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pmemcpy(void* d, const void* s, size_t n, unsigned nth) {
    std::vector<std::thread> th;
    const size_t per = ((n + nth - 1) / nth + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nth; ++t) {
        const size_t a = (size_t)t * per;
        if (a >= n) break;
        th.emplace_back([=]() { memcpy((char*)d + a, (const char*)s + a, a + per < n ? per : n - a); });
    }
    for (auto& t : th) t.join();
}
int main() {
    const size_t BLK = 128ull << 20, TOT = 1ull << 30, NB = TOT / BLK;
    void *d_in, *d_out, *h_in, *h_out;
    CK(hipMalloc(&d_in, TOT));
    CK(hipMalloc(&d_out, TOT));
    CK(hipHostMalloc(&h_in, TOT, hipHostMallocPortable));
    CK(hipHostMalloc(&h_out, TOT, hipHostMallocPortable));
    memset(h_in, 1, TOT);
    memset(h_out, 2, TOT);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto run = [&](bool up, bool down) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (size_t b = 0; b < NB; ++b) {
                if (up) CK(hipMemcpyAsync((char*)d_in + b * BLK, (char*)h_in + b * BLK, BLK, hipMemcpyHostToDevice, s1));
                if (down) CK(hipMemcpyAsync((char*)h_out + b * BLK, (char*)d_out + b * BLK, BLK, hipMemcpyDeviceToHost, s2));
            }
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        return best;
    };
    const double th2d = run(true, false), td2h = run(false, true), tdup = run(true, true);
    printf("{\"link_h2d_gbps\": %.2f, \"link_d2h_gbps\": %.2f, \"link_duplex_gbps_each_way\": %.2f, \"duplex_ms_per_gib_each_way\": %.2f", TOT / th2d / 1e9, TOT / td2h / 1e9,
           TOT / tdup / 1e9, tdup * 1e3);
    // pageable source
    char* pg = (char*)malloc(TOT);
    memset(pg, 3, TOT);
    {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now();
            CK(hipMemcpy(d_in, pg, TOT, hipMemcpyHostToDevice));
            best = std::min(best, now() - t0);
        }
        printf(", \"pageable_hipMemcpy_h2d_gbps\": %.2f", TOT / best / 1e9);
    }
    printf(", \"parallel_memcpy_to_pinned_gbps\": {");
    bool first = true;
    for (unsigned nth : {1u, 2u, 4u, 8u, 16u, 32u}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now();
            pmemcpy(h_in, pg, TOT, nth);
            best = std::min(best, now() - t0);
        }
        printf("%s\"%u\": %.2f", first ? "" : ", ", nth, TOT / best / 1e9);
        first = false;
    }
    printf("}");
    {
        const double t0 = now();
        CK(hipHostRegister(pg, TOT, hipHostRegisterDefault));
        const double treg = now() - t0;
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const double t1 = now();
            for (size_t b = 0; b < NB; ++b) CK(hipMemcpyAsync((char*)d_in + b * BLK, pg + b * BLK, BLK, hipMemcpyHostToDevice, s1));
            CK(hipStreamSynchronize(s1));
            best = std::min(best, now() - t1);
        }
        const double t2 = now();
        CK(hipHostUnregister(pg));
        printf(", \"hipHostRegister_ms_per_gib\": %.1f, \"hipHostUnregister_ms_per_gib\": %.1f, \"h2d_from_registered_gbps\": %.2f", treg * 1e3, (now() - t2) * 1e3, TOT / best / 1e9);
    }
    printf(", \"host_threads\": %u}\n", std::thread::hardware_concurrency());
    return 0;
}
