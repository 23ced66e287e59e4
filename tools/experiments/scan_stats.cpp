// Experiment (round 6): how much work phase D of tk_k_front has per tile, on the bench corpus -- pieces, certain starts, positions at which a
// piece MAY start ("stops"), the stops that are not certain, the certain starts whose next stop is not certain (listed for the scanners),
// and the lengths of the chains the scanners walk from them.  Host code over the product's host/device headers (like tests/hostsim).
//   g++ -O2 -std=c++17 -I. tools/experiments/scan_stats.cpp tiktoken_amd/csrc/tk_tables.cpp tiktoken_amd/csrc/tk_pattern.cpp tiktoken_amd/csrc/tk_regex.cpp -ldl -pthread -o /tmp/scan_stats
#include "../../tests/hostsim/tk_hostsim.cpp"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>

int main(int argc, char** argv) {
    const uint64_t n = (argc > 1 ? atoll(argv[1]) : 16) << 20;
    void* lib = dlopen(argc > 2 ? argv[2] : "tiktoken_amd/csrc/libtkcorpus.so", RTLD_NOW);
    auto gen = (int (*)(uint64_t, int, uint64_t, void*, void*, uint64_t, void*, int))dlsym(lib, "tkc_generate");
    std::vector<uint8_t> text(n + 128, 0);
    std::vector<uint64_t> off(n / 64 + 4);
    uint64_t nd = 0;
    gen(0x5EED0003ull, 1, n, text.data(), off.data(), n / 64 + 2, &nd, 8);
    // classes (o200k stock pattern: the tables need no vocabulary)
    TkTables T{};
    T.uc_stage1 = tk_uc_stage1;
    T.uc_stage2 = tk_uc_stage2;
    T.pattern = TK_PAT_O200K;
    T.pat = tk_stock_pat(TK_PAT_O200K);
    for (uint32_t a = 0; a < 16; ++a) T.cert[a] = (uint16_t)tk_certain_mask(TK_PAT_O200K, a);
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < nd; ++d)
        if (off[d] < n) brk[off[d] >> 5] |= 1u << (off[d] & 31);
    std::vector<uint8_t> cls2(n + 80, TK_C_END | 0x80);
    uint8_t last = TK_C_OT;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = tk_class_byte(T, text.data(), i, n, brk.data(), nullptr, nullptr);
        if ((c & 15u) == TK_C_CONT) cls2[i] = (uint8_t)(last | 0x40);
        else { cls2[i] = (uint8_t)c; last = (uint8_t)(c & 15u); }
    }
    PropAcc acc{cls2.data(), text.data(), n};
    const TkPat pat = T.pat;
    auto is_start = [&](uint64_t i) { return !(cls2[i] & 0x40u); };
    auto certain_at = [&](uint64_t i) -> bool {
        uint32_t c = cls2[i];
        if (c & 0x40u) return false;
        if (c & 0x80u) return true;
        if (i == 0) return false;
        return tk_certain_ctx(T, cls2[i - 1] & 15u, c & 15u, text.data(), i);
    };
    auto near = [&](uint64_t i) { return (i >= 2 && text[i - 2] == '\'') || (i >= 3 && text[i - 3] == '\''); };
    auto stop_at = [&](uint64_t i) -> bool {
        if (!is_start(i)) return false;
        if (certain_at(i)) return true;
        if (i == 0) return true;
        const uint32_t a = cls2[i - 1] & 15u, b = cls2[i] & 15u;
        return !(((tk_never_mask(TK_PAT_O200K, a) >> b) & 1u) && !near(i));
    };
    const uint64_t TILE = 3840;
    uint64_t tiles = 0, pieces = 0, certs = 0, stops = 0, unc = 0, listed = 0, evals = 0, sum_maxchain = 0, unc_reach = 0;
    std::map<uint32_t, uint64_t> chain_hist;
    std::map<std::string, uint64_t> kinds;
    uint64_t wave_iters_now = 0, wave_iters_dense = 0;
    for (uint64_t t0 = 0; t0 + TILE <= n; t0 += TILE, ++tiles) {
        const uint64_t t1 = t0 + TILE;
        std::vector<uint32_t> chains;
        uint64_t tile_unc = 0;
        for (uint64_t i = t0; i < t1; ++i) {
            if (!is_start(i)) continue;
            const bool c = certain_at(i), s = stop_at(i);
            stops += s;
            if (s && !c) { ++unc; ++tile_unc; }
            if (!c) continue;
            ++certs;
            // next stop
            uint64_t j = i + 1;
            while (j < n && !stop_at(j)) ++j;
            if (j < n && certain_at(j)) continue;  // round 0 settles it
            ++listed;
            {   // what kind of start is it?  (class at the start, class at the first stop behind it, is the piece's end the SECOND stop and is that one certain)
                uint64_t j2 = j + 1;
                while (j2 < n && !stop_at(j2)) ++j2;
                uint64_t e1 = tk_piece_end(acc, i, pat);
                const bool second = e1 == j2, second_cert = j2 < n && certain_at(j2);
                const bool adjacent = j == tk_next_char(acc, i);
                char key[64];
                snprintf(key, sizeof key, "%2u->%2u adj%d end@2nd%d cert2nd%d", cls2[i] & 15u, cls2[j] & 15u, (int)adjacent, (int)second, (int)second_cert);
                kinds[key]++;
            }
            uint32_t len = 0;
            uint64_t q = i;
            for (;;) {
                uint64_t e = tk_piece_end(acc, q, pat);
                if (e <= q) e = tk_next_char(acc, q);
                ++len;
                if (e >= t1 || certain_at(e)) break;
                q = e;
            }
            chains.push_back(len);
            evals += len;
            chain_hist[len > 12 ? 12 : len]++;
        }
        // wave-iterations of the scanner: now = lanes in list order, 64 per wavefront, each wavefront as long as its longest chain
        for (size_t b = 0; b < chains.size(); b += 64) {
            uint32_t mx = 0;
            for (size_t k = b; k < chains.size() && k < b + 64; ++k) mx = chains[k] > mx ? chains[k] : mx;
            wave_iters_now += mx;
            sum_maxchain += mx;
        }
        wave_iters_dense += (chains.size() + tile_unc + 63) / 64;
    }
    // why tiles are deferred to the workgroup-wide scanner: (a) no certain start in the 128 bytes of left context while the tile's first char is not one;
    // (b) a piece that starts in the tile (or covers its start) ends more than 128 bytes behind the tile's end
    {
        uint64_t walk = 0, leave = 0, both = 0;
        std::vector<uint8_t> is_piece_start(n + 1, 0);
        for (uint64_t qq = 0; qq < n;) { is_piece_start[qq] = 1; uint64_t e = tk_piece_end(acc, qq, pat); if (e <= qq) e = tk_next_char(acc, qq); qq = e; }
        for (uint64_t t0 = TILE; t0 + TILE <= n; t0 += TILE) {
            uint64_t f = t0;
            while (f < t0 + TILE && !is_start(f)) ++f;
            bool w = false, l = false;
            if (!certain_at(f)) {
                w = true;
                for (uint64_t j = t0 - 1; j + 128 > t0 && j > 0; --j)
                    if (certain_at(j)) { w = false; break; }
            }
            // the last piece that starts before the tile's end: where does it end?
            uint64_t ls = t0 + TILE - 1;
            while (ls > t0 && !is_piece_start[ls]) --ls;
            uint64_t e = tk_piece_end(acc, ls, pat);
            if (e > t0 + TILE + 128 - 8) l = true;
            walk += w; leave += l; both += w && l;
        }
        printf("deferred tiles (of %llu): no certain start in the left context %.2f %%, last piece leaves the window %.2f %%, both %.2f %%\n", (unsigned long long)tiles,
               100.0 * walk / tiles, 100.0 * leave / tiles, 100.0 * both / tiles);
    }
    // pieces
    uint64_t q = 0;
    while (q < tiles * TILE) { uint64_t e = tk_piece_end(acc, q, pat); if (e <= q) e = tk_next_char(acc, q); ++pieces; q = e; }
    printf("tiles %llu  per tile: pieces %.1f certain starts %.1f stops %.1f uncertain stops %.1f listed %.1f evaluations %.1f\n", (unsigned long long)tiles,
           (double)pieces / tiles, (double)certs / tiles, (double)stops / tiles, (double)unc / tiles, (double)listed / tiles, (double)evals / tiles);
    printf("scanner wave-iterations per tile: now %.2f (a wavefront runs as long as its longest chain); all listed + all uncertain stops evaluated once, dense: %.2f\n",
           (double)wave_iters_now / tiles, (double)wave_iters_dense / tiles);
    printf("chain length histogram (evaluations per listed start): ");
    for (auto& kv : chain_hist) printf("%u:%.3f ", kv.first, (double)kv.second / listed);
    printf("\n");
    std::vector<std::pair<uint64_t, std::string>> kv;
    for (auto& k : kinds) kv.push_back({k.second, k.first});
    std::sort(kv.rbegin(), kv.rend());
    for (size_t i = 0; i < kv.size() && i < 25; ++i) printf("  %-44s %.3f per tile  (%.1f %% of the listed)\n", kv[i].second.c_str(), (double)kv[i].first / tiles, 100.0 * kv[i].first / listed);
    return 0;
}
