// The generic pat_str engine under AddressSanitizer / UBSan: the compiler (tk_regex.cpp) on well- and ill-formed patterns, and the lane
// code of the two split kernels (tk_regex_split.h) on random text -- valid UTF-8, truncated chars, stray continuation bytes, NULs -- in
// buffers sized exactly as the device's (n bytes + 64 readable bytes; bitmaps of (n + 31) / 32 + 2 words).  Test infrastructure only
// (tests/test_regex_engine.py builds and runs it); prints "ok <patterns> <splits>" or dies in the sanitizer.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <vector>

#include "../../tiktoken_amd/csrc/tk_regex_host.h"
#include "../../tiktoken_amd/csrc/tk_regex_split.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}

static std::string good_alt(int depth, int n);
static std::string random_pattern() {
    static const char* const frag[] = {"a", "b", "\\s", "\\S", "\\d", "\\w", "\\p{L}", "\\p{Lu}", "\\P{N}", "[a-c]", "[^a\\s]", "[^\\S\\n]", ".", "é", "中", "\\n", "\\x41", "\\u4e2d",
                                       "(", ")", "(?:", "(?i:", "(?>", "(?=", "(?!", "(?s:", "|", "|", "?", "*", "+", "{1,3}", "{2}", "{2,}", "??", "*+", "+?", "^", "$", "\\z",
                                       "[", "]", "\\", "{", "}", "(?<=", "\\b", "\\1", "\\p{Han}", "[[:alpha:]]", "&&", "(?m)", "(?i)", "\\xZ", "\\u{110000}", "\xC3", "\x80",
                                       "[z-a]", "{3,2}", "\\p{", "(?P<n>", "(?<n>", "-", "[a-", "[^]", "\\Q", "\\E", "'s", " ?"};
    if (rnd() % 3) return good_alt(0, 1 + rnd() % 4) + (rnd() % 4 ? "|[\\s\\S]" : "") + (rnd() % 4 ? "" : "|\\s+$");
    std::string s;
    const uint32_t k = 1 + rnd() % 14;
    for (uint32_t i = 0; i < k; ++i) s += frag[rnd() % (sizeof frag / sizeof frag[0])];
    if (rnd() % 3) s += "|[\\s\\S]";
    return s;
}

// a well-formed pattern of the supported syntax (the Python tests compare such patterns with `regex`; here they only have to stay in bounds)
static std::string good_atom(int depth);
static std::string good_alt(int depth, int n) {
    static const char* const q[] = {"", "", "", "?", "*", "+", "{1,3}", "{2}", "{2,}", "??", "*?", "+?", "?+", "*+", "++"};
    std::string s;
    for (int a = 0; a < n; ++a) {
        if (a) s += "|";
        const int parts = 1 + rnd() % 3;
        for (int k = 0; k < parts; ++k) s += good_atom(depth) + (k == 0 ? (rnd() % 2 ? "" : "+") : q[rnd() % 15]);
        if (rnd() % 5 == 0) s += std::string(rnd() % 2 ? "(?=" : "(?!") + good_atom(depth + 1) + ")";
    }
    return s;
}
static std::string good_atom(int depth) {
    static const char* const a[] = {"a", "b", " ", "\\n", "'", "s", "k", "é", "中", "[a-c]", "[^a\\s]", "\\s", "\\S", "\\d", "\\w", "\\p{L}", "\\p{Lu}", "\\P{N}", "[\\s\\S]",
                                    "[^\\S\\n]", "[x1\\p{Ll}]", "[^\\r\\n\\p{L}\\p{N}]", ".", "\\x61", "[\\x{4e00}-\\x{9fff}]", "\\p{M}", "\\p{Nd}",
                                    "\\p{Han}", "[\\p{L}&&[^a-c\\p{Han}]]", "[\\w--\\d]", "a\\b", "\\B.", "(?<=\\s)a", "(?<!ab|\\p{Lu})\\w", "(?m:^a|b$)"};
    if (depth > 2 || rnd() % 4) return a[rnd() % (sizeof a / sizeof a[0])];
    static const char* const open[] = {"(?:", "(?:", "(", "(?>", "(?s:", "(?i:"};
    const char* o = open[rnd() % 6];
    if (!strcmp(o, "(?i:")) return std::string(o) + (rnd() % 2 ? "s|k|ab" : "'s|'ll") + ")";
    return std::string(o) + good_alt(depth + 1, 1 + rnd() % 3) + ")";
}

static std::vector<uint8_t> random_text(uint32_t n) {
    static const char* const unit[] = {"a", "b", "c", " ", "\n", "A", "1", "'", ".", "é", "中", "😀", "\xE4\xB8", "\x80", "\xF0\x9F", "\xFF", "\xC3", "ab", "  ", "\r\n", "s", "K"};
    std::vector<uint8_t> t;
    while (t.size() < n) {
        const char* u = unit[rnd() % (sizeof unit / sizeof unit[0])];
        uint32_t rep = (rnd() % 16 == 0) ? 1 + rnd() % 600 : 1;
        if (rnd() % 64 == 0) {
            t.push_back(0);
            continue;
        }
        while (rep-- && t.size() < n) t.insert(t.end(), (const uint8_t*)u, (const uint8_t*)u + strlen(u));
    }
    t.resize(n);
    return t;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    uint64_t compiled = 0, splits = 0, with_dfa = 0;
    for (int r = 0; r < rounds; ++r) {
        const std::string pat = random_pattern();
        TkRxCompiled c;
        if (!tk_rx_compile(pat.c_str(), &c).empty()) continue;
        ++compiled;
        const TkRxProg P = c.view();
        for (int k = 0; k < 6; ++k) {
            static const uint32_t sizes[] = {0, 1, 7, 255, 256, 257, 1023, 1025, 5000, 40000};
            const uint32_t n = sizes[rnd() % 10];
            std::vector<uint8_t> body = random_text(n);
            // exactly the device's sizes: no slack that would hide an overrun
            uint8_t* text = (uint8_t*)malloc(n + 64);
            if (n) memcpy(text, body.data(), n);
            memset(text + n, 0, 64);
            const uint64_t nw = ((uint64_t)n + 31) / 32 + 2;
            uint32_t *brk = (uint32_t*)calloc(nw, 4), *spec = (uint32_t*)calloc(nw, 4), *gst = (uint32_t*)calloc(nw, 4);
            std::vector<uint32_t> doc = {0};
            while (doc.back() < n) {
                const uint64_t nx = (uint64_t)doc.back() + 1 + rnd() % (n / 3 + 1);
                doc.push_back(nx > n ? n : (uint32_t)nx);
            }
            for (uint32_t d : doc)
                if (d < n) brk[d >> 5] |= 1u << (d & 31);
            // special tokens at random places (every second text): start bit, interior bits, hard edges -- as tk_k_spec_resolve leaves them
            uint32_t *ss = (uint32_t*)calloc(nw, 4), *si = (uint32_t*)calloc(nw, 4);
            const bool with_specials = (k & 1) && n > 8;
            if (with_specials)
                for (uint32_t at = rnd() % 40; at + 1 < n; at += 1 + rnd() % 700) {
                    uint32_t len = 1 + rnd() % 90;
                    if (at + len > n) len = n - at;
                    ss[at >> 5] |= 1u << (at & 31);
                    brk[at >> 5] |= 1u << (at & 31);
                    for (uint32_t j = at + 1; j < at + len; ++j) si[j >> 5] |= 1u << (j & 31);
                    if (at + len < n) brk[(at + len) >> 5] |= 1u << ((at + len) & 31);
                    at += len;
                }
            auto run_lanes = [&](auto dfa_tag, uint32_t shift) {  // (the matcher: the program, or the pattern's DFA where it has one)
                constexpr int DFA = decltype(dfa_tag)::value;  // (TK_RX_M_PROGRAM, _DFA, _DFA_PREV)
                const uint32_t nseg = (uint32_t)(((uint64_t)n + (1u << shift) - 1) >> shift);
                uint32_t* xexit = (uint32_t*)malloc((nseg + 2) * 4);
                uint32_t* lmerge = (uint32_t*)malloc((nseg + 2) * 4);
                uint32_t* lexit = (uint32_t*)malloc((nseg + 2) * 4);
                memset(spec, 0, nw * 4);
                uint32_t* sgap = (uint32_t*)calloc(nw, 4);
                uint32_t* lnk = (uint32_t*)calloc(nw, 4);
                uint32_t* lgap = (uint32_t*)calloc(nw, 4);
                TkRxText t{text, n, brk, with_specials ? ss : nullptr, with_specials ? si : nullptr, 0xFFFFFFFFu, false};
                t.ahead = (rnd() % 3 == 0) ? 64u : (DFA ? TK_RX_AHEAD_DFA : TK_RX_AHEAD);  // (bytes a speculative match may look beyond its segment)
                if constexpr (DFA) {  // (the device's form of the DFA's speculative lane: one loop over the segment)
                    for (uint32_t s = 0; s < nseg; ++s) tk_rx_speculate_lane_flat<DFA == TK_RX_M_DFA_PREV>(P, t, s, shift, spec, sgap, xexit);
                } else {
                    for (uint32_t s = 0; s < nseg; ++s) tk_rx_speculate_lane<DFA>(P, t, s, shift, spec, sgap, xexit);
                }
                for (uint32_t s = 0; s < nseg; ++s) tk_rx_link_lane<DFA>(P, t, s, shift, spec, xexit, lnk, lgap, lmerge, lexit);
                const TkRxMaps M{spec, sgap, xexit, lnk, lgap, lmerge, lexit, shift};
                for (int by_group = 0; by_group < 2; ++by_group) {  // (one lane per document; a group of lanes per document)
                    memset(gst, 0, nw * 4);
                    for (size_t d = 0; d + 1 < doc.size(); ++d) {
                        uint32_t err_pos = 0;
                        auto orb = [&](uint32_t w, uint32_t bits, uint32_t gaps) {
                            if (w >= nw || (gaps & ~(gst[w] | bits))) abort();  // (a gap char is a start)
                            gst[w] |= bits;
                        };
                        if (by_group) (void)tk_rx_resolve_group_host<DFA>(P, t, M, doc[d], doc[d + 1], orb, &err_pos);
                        else (void)tk_rx_resolve_lane<DFA>(P, t, M, doc[d], doc[d + 1], orb, &err_pos);
                    }
                }
                free(lmerge);
                free(lexit);
                free(lnk);
                free(lgap);
                ++splits;
                free(xexit);
                free(sgap);
            };
            for (uint32_t shift : {TK_RX_SEG_SHIFT_SMALL, TK_RX_SEG_SHIFT_LARGE}) {
                run_lanes(std::integral_constant<int, TK_RX_M_PROGRAM>{}, shift);
                if (c.has_dfa() && (c.dfa_flags & 1u)) run_lanes(std::integral_constant<int, TK_RX_M_DFA_PREV>{}, shift), ++with_dfa;
                else if (c.has_dfa()) run_lanes(std::integral_constant<int, TK_RX_M_DFA>{}, shift), ++with_dfa;
            }
            free(text);
            free(brk);
            free(spec);
            free(gst);
            free(ss);
            free(si);
        }
    }
    printf("ok %llu %llu (%llu through a DFA)\n", (unsigned long long)compiled, (unsigned long long)splits, (unsigned long long)with_dfa);
    return 0;
}
