// tk_compile_pattern (the family parser of tk_pattern.cpp first, then the generic compiler) on random, mostly ill-formed pattern strings
// and on mutations of the stock patterns, under AddressSanitizer / UBSan: whatever a caller hands to tk_create must be refused or compiled,
// never crash.  argv[1..]: stock patterns to mutate.  Test infrastructure only (tests/test_patterns.py builds and runs it).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../tiktoken_amd/csrc/tk_tables.h"

static uint64_t rs = 0xD1B54A32D192ED03ull;
static uint32_t rnd() {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (uint32_t)(rs >> 11);
}

int main(int argc, char** argv) {
    static const char* const frag[] = {"'s", "'(?i:[sdmt]|ll|ve|re)", "(?i:'s|'t)", "|", "|", " ?", "\\p{L}+", "\\p{N}{1,3}", "\\p{N}", "[^\\s\\p{L}\\p{N}]+", "[\\r\\n]*", "\\s*[\\r\\n]+",
                                       "\\s+(?!\\S)", "\\s+", "\\s", "$", "++", "?+", "*+", "[", "]", "(", ")", "(?", "(?i", "{", "}", "{1,", "\\", "\\p{", "\\p{L", "[^", "[^\\r\\n\\p{L}\\p{N}]?",
                                       "[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*", "[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+", "/", "\xC3", "\xE4\xB8\xAD", "a", "|'", "'", "(?i:", "(?:", "{1,999999}", "{0}", "\\s++$"};
    uint64_t n = 0, ok = 0;
    const int rounds = getenv("TK_SAN_ROUNDS") ? atoi(getenv("TK_SAN_ROUNDS")) : 20000;
    for (int r = 0; r < rounds; ++r) {
        std::string p;
        if (argc > 1 && rnd() % 2) {  // a stock pattern with a few bytes deleted, doubled or replaced
            p = argv[1 + rnd() % (argc - 1)];
            for (int k = rnd() % 4; k >= 0 && !p.empty(); --k) {
                const size_t at = rnd() % p.size();
                switch (rnd() % 3) {
                    case 0: p.erase(at, 1 + rnd() % 3); break;
                    case 1: p.insert(at, p.substr(at, 1 + rnd() % 4)); break;
                    default: p[at] = "|()[]{}\\?+*^$'sS "[rnd() % 18]; break;
                }
            }
        } else {
            for (int k = 1 + rnd() % 12; k > 0; --k) p += frag[rnd() % (sizeof frag / sizeof frag[0])];
        }
        for (size_t i = 0; i < p.size(); ++i)
            if (!p[i]) p[i] = 'x';
        TkPat pat;
        uint16_t cert[16];
        TkRxCompiled rx;
        const std::string e = tk_compile_pattern(p.c_str(), &pat, (r & 7) ? nullptr : cert, &rx);  // (deriving the certain starts takes 30 ms: one in eight)
        ++n;
        ok += e.empty();
    }
    // `.tiktoken` text (base64 token, blank, rank per line: reference tiktoken/load.py:159-171), damaged in every way; and vocabularies that
    // the table builder has to refuse (a missing byte, duplicate tokens, duplicate ranks, ranks beyond 2^31) or accept (sparse ranks)
    uint64_t parsed = 0, built = 0;
    {
        static const char b64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        std::string good;
        for (int b = 0; b < 256; ++b) {
            const uint32_t v = (uint32_t)b << 16;
            good += b64[(v >> 18) & 63];
            good += b64[(v >> 12) & 63];
            good += "== ";
            good += std::to_string(b);
            good += "\n";
        }
        good += "YWI= 256\nYWJj 257\n";
        for (int r = 0; r < rounds / 4; ++r) {
            std::string t = good;
            for (int k = rnd() % 6; k > 0 && !t.empty(); --k) {
                const size_t at = rnd() % t.size();
                switch (rnd() % 4) {
                    case 0: t.erase(at, 1 + rnd() % 5); break;
                    case 1: t.insert(at, t.substr(rnd() % t.size(), rnd() % 9)); break;
                    case 2: t[at] = (char)rnd(); break;
                    default: t.insert(at, std::string(1 + rnd() % 3, "= \n9A"[rnd() % 6])); break;
                }
            }
            std::vector<uint8_t> blob;
            std::vector<uint64_t> off;
            std::vector<uint32_t> ids;
            const std::string e = tk_parse_tiktoken((const uint8_t*)t.data(), t.size(), &blob, &off, &ids);
            if (!e.empty()) continue;
            ++parsed;
            TkHostTables H;
            uint64_t soff = 0;
            if (tk_build_tables(blob.data(), off.data(), ids.data(), ids.size(), blob.data(), &soff, ids.data(), 0, argv[1 + rnd() % (argc - 1)], &H).empty()) ++built;
        }
        for (int r = 0; r < 200; ++r) {  // hand-made vocabularies
            std::vector<uint8_t> blob;
            std::vector<uint64_t> off = {0};
            std::vector<uint32_t> ids;
            const int kind = r % 5;
            for (int b = 0; b < 256; ++b) {
                if (kind == 1 && b == (int)(rnd() % 256)) continue;  // a byte is missing
                blob.push_back((uint8_t)b);
                off.push_back(blob.size());
                ids.push_back(kind == 4 ? (uint32_t)b * 1000u : (uint32_t)b);  // (4: sparse ranks)
            }
            const int extra = rnd() % 40;
            for (int k = 0; k < extra; ++k) {
                const int len = 2 + rnd() % 30;
                for (int j = 0; j < len; ++j) blob.push_back((uint8_t)("abcde "[rnd() % 6]));
                off.push_back(blob.size());
                ids.push_back(kind == 2 ? 5u : (kind == 3 ? 0x80000000u + k : (kind == 4 ? 300000u + 7u * k : 256u + k)));  // 2: duplicate ranks, 3: too large
            }
            TkHostTables H;
            uint64_t soff = 0;
            (void)tk_build_tables(blob.data(), off.data(), ids.data(), ids.size(), blob.data(), &soff, ids.data(), 0, argv[1], &H);
        }
    }
    printf("ok %llu %llu %llu %llu\n", (unsigned long long)n, (unsigned long long)ok, (unsigned long long)parsed, (unsigned long long)built);
    return 0;
}
