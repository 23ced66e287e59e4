// tk_compile_pattern (the family parser of tk_pattern.cpp first, then the generic compiler) on random, mostly ill-formed pattern strings
// and on mutations of the stock patterns, under AddressSanitizer / UBSan: whatever a caller hands to tk_create must be refused or compiled,
// never crash.  argv[1..]: stock patterns to mutate.  Test infrastructure only (tests/test_patterns.py builds and runs it).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

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
    printf("ok %llu %llu\n", (unsigned long long)n, (unsigned long long)ok);
    return 0;
}
