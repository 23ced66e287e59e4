// Experiment (round 6): the "sync point" rule of tk_k_front (tk_fused.h, TKF_SYNC_POINTS) checked on the bench corpus -- every position that qualifies, not only
// the ones the kernel uses: the scan from it must end where its piece ends.  Host code over the product headers (like tests/hostsim).
//   g++ -O2 -std=c++17 -I. tools/experiments/sync_points.cpp tiktoken_amd/csrc/tk_tables.cpp tiktoken_amd/csrc/tk_pattern.cpp tiktoken_amd/csrc/tk_regex.cpp -ldl -pthread -o /tmp/sync && /tmp/sync 2
#include "../../tests/hostsim/tk_hostsim.cpp"
#include <dlfcn.h>
#include <stdio.h>
#include <map>
int main(int argc, char** argv) {
    const uint64_t n = 64ull << 20;
    const int patid = argc > 1 ? atoi(argv[1]) : TK_PAT_O200K;
    void* lib = dlopen("tiktoken_amd/csrc/libtkcorpus.so", RTLD_NOW);
    auto gen = (int (*)(uint64_t, int, uint64_t, void*, void*, uint64_t, void*, int))dlsym(lib, "tkc_generate");
    std::vector<uint8_t> text(n + 128, 0); std::vector<uint64_t> off(n / 64 + 4); uint64_t nd = 0;
    gen(argc > 3 ? 0x5EED0002ull : 0x5EED0003ull, argc > 2 ? atoi(argv[2]) : 1, n, text.data(), off.data(), n / 64 + 2, &nd, 8);
    TkTables T{}; T.uc_stage1 = tk_uc_stage1; T.uc_stage2 = tk_uc_stage2; T.pattern = patid; T.pat = tk_stock_pat(patid);
    for (uint32_t a = 0; a < 16; ++a) T.cert[a] = (uint16_t)tk_certain_mask(patid, a);
    std::vector<uint32_t> brk((n + 31) / 32 + 2, 0);
    for (uint64_t d = 0; d < nd; ++d) if (off[d] < n) brk[off[d] >> 5] |= 1u << (off[d] & 31);
    std::vector<uint8_t> cls2(n + 80, TK_C_END | 0x80); uint8_t last = TK_C_OT;
    for (uint64_t i = 0; i < n; ++i) { uint32_t c = tk_class_byte(T, text.data(), i, n, brk.data(), nullptr, nullptr);
        if ((c & 15u) == TK_C_CONT) cls2[i] = (uint8_t)(last | 0x40); else { cls2[i] = (uint8_t)c; last = (uint8_t)(c & 15u); } }
    PropAcc acc{cls2.data(), text.data(), n};
    std::vector<uint8_t> truth(n + 1, 0);
    for (uint64_t q = 0; q < n;) { truth[q] = 1; uint64_t e = tk_piece_end(acc, q, T.pat); if (e <= q) e = tk_next_char(acc, q); q = e; }
    truth[n] = 1;
    auto certain_at = [&](uint64_t i) -> bool { uint32_t c = cls2[i]; if (c & 0x40u) return false; if (c & 0x80u) return true; if (i == 0) return false;
        return tk_certain_ctx(T, cls2[i - 1] & 15u, c & 15u, text.data(), i); };
    auto near = [&](uint64_t i) { return (i >= 1 && text[i-1] == '\'') || (i >= 2 && text[i - 2] == '\'') || (i >= 3 && text[i - 3] == '\''); };
    // sync points: kind 1 = LL after LL/LC; kind 2 = LC after LL/LC whose {LC,MK} run ends (inside 256 bytes) with a char that is not LU
    auto sync_kind = [&](uint64_t s) -> int {
        if (s == 0 || (cls2[s] & 0xC0)) return 0;
        int c = cls2[s] & 15, p = cls2[s - 1] & 15;
        if (patid == TK_PAT_CL100K) {  // any letter behind a letter: inside `\p{L}++`
            const bool lp = p == TK_C_LU || p == TK_C_LL || p == TK_C_LC, lc = c == TK_C_LU || c == TK_C_LL || c == TK_C_LC;
            return (lp && lc && !(cls2[s - 1] & 0x80) && !near(s)) ? 1 : 0;
        }
        if (!(p == TK_C_LL || p == TK_C_LC) || (cls2[s-1] & 0x80)) return 0;
        if (near(s)) return 0;
        if (patid == TK_PAT_CL100K) return 0;
        if (c == TK_C_LL) return 1;
        if (c == TK_C_LC) { uint64_t j = s; while (j < n && j < s + 200 && ((cls2[j] & 15) == TK_C_LC || (cls2[j] & 15) == TK_C_MK) && !(cls2[j] & 0x80)) ++j;
            if (j >= s + 200) return 0; int x = cls2[j] & 15; if (x == TK_C_LU && !(cls2[j] & 0x80)) return 0; return 2; }
        return 0; };
    uint64_t cnt[3] = {0,0,0}, badc[3] = {0,0,0};
    for (uint64_t s = 1; s < n; s += 1) { int k = sync_kind(s); if (!k) continue; ++cnt[k];
        if (truth[s]) { ++badc[k]; if (badc[k] < 5) printf("kind %d IS A START at %llu [%.*s]\n", k, (unsigned long long)s, 24, &text[s-8]); continue; }
        uint64_t e = tk_piece_end(acc, s, T.pat); uint64_t te = s + 1; while (!truth[te]) ++te;
        if (e != te) { ++badc[k]; if (badc[k] < 5) printf("kind %d END differs at %llu: %llu vs %llu [%.*s]\n", k, (unsigned long long)s, (unsigned long long)e, (unsigned long long)te, 40, &text[s-8]); } }
    printf("pat %d: sync LL %llu (bad %llu), sync LC %llu (bad %llu)\n", patid, (unsigned long long)cnt[1], (unsigned long long)badc[1], (unsigned long long)cnt[2], (unsigned long long)badc[2]);
    const uint64_t TILE = 3840; uint64_t tiles = 0, walk = 0, cov1 = 0, cov2 = 0;
    for (uint64_t t0 = TILE; t0 + TILE <= n; t0 += TILE, ++tiles) {
        uint64_t f = t0; while (f < t0 + TILE && (cls2[f] & 0x40)) ++f;
        if (certain_at(f)) continue;
        bool w = true; for (uint64_t j = t0 - 1; j + 128 > t0; --j) if (certain_at(j)) { w = false; break; }
        if (!w) continue; ++walk;
        int best = 0; for (uint64_t j = t0 - 127; j <= f; ++j) { int k = sync_kind(j); if (k == 1) best = 1; else if (k == 2 && !best) best = 2; }
        cov1 += best == 1; cov2 += best == 2;
    }
    printf("tiles %llu walk %llu covered by LL sync %llu, by LC sync %llu\n", (unsigned long long)tiles, (unsigned long long)walk, (unsigned long long)cov1, (unsigned long long)cov2);
}
