// Experiment (round 6): the "letter run that leaves the window" rule of tk_k_front (tk_fused.h, TKF_EXTEND) checked on the bench corpora: for every
// tile whose last piece ends behind the window, does the rule apply, and does it give the piece's end?  Host code over the product headers.
//   g++ -O2 -std=c++17 -I. tools/experiments/extend_piece.cpp tiktoken_amd/csrc/tk_tables.cpp tiktoken_amd/csrc/tk_pattern.cpp tiktoken_amd/csrc/tk_regex.cpp -ldl -pthread -o /tmp/extend && /tmp/extend 2
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
    const bool CL = patid == TK_PAT_CL100K;
    auto in_set = [&](int c) { return CL ? (c == TK_C_LU || c == TK_C_LL || c == TK_C_LC) : (c == TK_C_LC || c == TK_C_MK); };
    auto letter = [&](int c) { return c == TK_C_LU || c == TK_C_LL || c == TK_C_LC || (!CL && c == TK_C_MK); };
    auto prefix = [&](int c) { return c == TK_C_SP || c == TK_C_WSO || c == TK_C_SL || c == TK_C_OT || (CL && c == TK_C_MK); };
    const uint64_t TILE = 3840; uint64_t tiles = 0, leave = 0, applies = 0, right = 0, wrong = 0, toolong = 0;
    for (uint64_t t0 = TILE; t0 + TILE + 4096 <= n; t0 += TILE, ++tiles) {
        const uint64_t wend = t0 + TILE + 128;  // first position behind the window
        // pieces that start before the tile's end and end behind the window: the last such
        uint64_t ls = t0 + TILE - 1; while (!truth[ls]) --ls;
        uint64_t e = ls + 1; while (!truth[e]) ++e;
        if (e <= wend) continue;  // (the kernel's condition is a look at position wend or beyond: e >= wend after the last char; close enough for a count)
        ++leave;
        int cp = cls2[ls] & 15; bool hardin = false;
        uint64_t nx = ls + 1; while (cls2[nx] & 0x40) ++nx;
        bool okp = letter(cp) || (prefix(cp) && letter(cls2[nx] & 15) && !(cls2[nx] & 0x80));
        uint64_t lastc = wend - 1; while (cls2[lastc] & 0x40) --lastc;
        // o200k: the run may go on with lower-case letters (LL) -- from the first of them on the matcher is in the alternative's lower-case part, where an
        // upper-case letter ENDS the piece; before it (state unknown: LC and MK are in both parts) an upper-case letter lets the piece go on: not decided here
        auto in_run = [&](int c) { return in_set(c) || (!CL && c == TK_C_LL); };
        if (!okp || !in_run(cls2[lastc] & 15)) continue;
        bool seen_ll = !CL && (cls2[lastc] & 15) == TK_C_LL;
        uint64_t x = wend; while (x < n && (cls2[x] & 0x40)) ++x;  // (the char that straddles the window's end belongs to the last char)
        while (x < n && x < wend + 2048 && in_run(cls2[x] & 15) && !(cls2[x] & 0x80)) { seen_ll = seen_ll || (cls2[x] & 15) == TK_C_LL; ++x; while (x < n && (cls2[x] & 0x40)) ++x; }
        if (x >= wend + 2048) { ++toolong; continue; }
        int cx = cls2[x] & 15;
        if (!CL && !(cls2[x] & 0x80) && (cx == TK_C_AP || (cx == TK_C_LU && !seen_ll))) continue;
        ++applies; (void)hardin;
        if (x == e) ++right; else { ++wrong; if (wrong < 6) printf("WRONG tile at %llu: rule %llu true %llu\n", (unsigned long long)t0, (unsigned long long)x, (unsigned long long)e); }
    }
    printf("pat %d: tiles %llu, last piece leaves the window %llu (%.2f %%); the rule applies to %llu (right %llu, wrong %llu), run longer than 2 KiB %llu\n", patid,
           (unsigned long long)tiles, (unsigned long long)leave, 100.0 * leave / tiles, (unsigned long long)applies, (unsigned long long)right, (unsigned long long)wrong, (unsigned long long)toolong);
}
