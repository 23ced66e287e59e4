// The hand-written scanners and the 16-bytes-per-lane classification of the device headers (driven by tk_hostsim.cpp) under
// AddressSanitizer / UBSan, on random text -- valid UTF-8, truncated chars, stray continuation bytes, NULs, long runs -- with random document
// starts.  argv[1..]: pat_str of the encodings to run.  Test infrastructure only (tests/test_device_logic_sim.py builds and runs it).
#include "tk_hostsim.cpp"

#include <stdio.h>
#include <stdlib.h>

static uint64_t rs = 0x2545F4914F6CDD1Dull;
static uint32_t rnd() {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (uint32_t)(rs >> 11);
}

int main(int argc, char** argv) {
    const int rounds = getenv("TK_SAN_ROUNDS") ? atoi(getenv("TK_SAN_ROUNDS")) : 60;
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off = {0};
    std::vector<uint32_t> ids;
    for (int b = 0; b < 256; ++b) {
        blob.push_back((uint8_t)b);
        off.push_back(blob.size());
        ids.push_back((uint32_t)b);
    }
    static const char* const unit[] = {"a", "B", "c", " ", "\n", "\r\n", "1", "'", "'s", "'LL", ".", "/", "é", "中", "😀", "\xE4\xB8", "\x80", "\xF0\x9F", "\xFF", "\xC3", "ab", "  ", "\t", "ſ", "K", "٣"};
    uint64_t runs = 0;
    for (int a = 1; a < argc; ++a) {
        char err[256];
        uint64_t soff = 0;
        void* sim = tks_create(blob.data(), off.data(), ids.data(), 256, blob.data(), &soff, ids.data(), 0, argv[a], err, sizeof err);
        if (!sim) {
            fprintf(stderr, "tks_create: %s\n", err);
            return 2;
        }
        for (int r = 0; r < rounds; ++r) {
            static const uint32_t sizes[] = {0, 1, 15, 16, 17, 63, 64, 65, 200, 4095, 4096, 4097, 9000, 20000};
            const uint32_t n = sizes[rnd() % 14];
            std::vector<uint8_t> t;
            while (t.size() < n) {
                const char* u = unit[rnd() % (sizeof unit / sizeof unit[0])];
                uint32_t rep = (rnd() % 12 == 0) ? 1 + rnd() % 5000 : 1;
                if (rnd() % 64 == 0) {
                    t.push_back(0);
                    continue;
                }
                while (rep-- && t.size() < n) t.insert(t.end(), (const uint8_t*)u, (const uint8_t*)u + strlen(u));
            }
            t.resize(n);
            t.shrink_to_fit();
            std::vector<uint64_t> doc = {0};
            while (doc.back() < n) {
                const uint64_t nx = doc.back() + 1 + rnd() % (n / 2 + 1);
                doc.push_back(nx > n ? n : nx);
            }
            std::vector<uint8_t> starts(n + 1);
            const uint8_t* tp = n ? t.data() : (const uint8_t*)"";
            tks_pretok(sim, tp, n, doc.data(), doc.size() - 1, starts.data());
            tks_pretok_bits(sim, tp, n, doc.data(), doc.size() - 1, starts.data());
            tks_pretok_tiles(sim, tp, n, doc.data(), doc.size() - 1, starts.data(), 3840, 128);
            tks_pretok_tiles(sim, tp, n, doc.data(), doc.size() - 1, starts.data(), 64, 16);
            uint64_t first_bad = 0;
            uint32_t what = 0;
            tks_chunk_check(sim, tp, n, doc.data(), doc.size() - 1, nullptr, nullptr, 3840, 128, 4096, &first_bad, &what);
            runs += 5;
        }
        tks_destroy(sim);
    }
    // With a real vocabulary (TK_SAN_VOCAB: a .tiktoken text file): the whole-piece probes of the three tables and the per-lane merge on
    // random pieces -- vocabulary tokens, their prefixes, concatenations, random bytes -- placed at the end of an n + 64 byte buffer.
    if (const char* vp = getenv("TK_SAN_VOCAB")) {
        FILE* f = fopen(vp, "rb");
        if (!f) {
            fprintf(stderr, "cannot open %s\n", vp);
            return 2;
        }
        std::vector<uint8_t> raw;
        uint8_t buf[65536];
        for (size_t k; (k = fread(buf, 1, sizeof buf, f)) > 0;) raw.insert(raw.end(), buf, buf + k);
        fclose(f);
        std::vector<uint8_t> vb;
        std::vector<uint64_t> vo;
        std::vector<uint32_t> vi;
        const std::string e = tk_parse_tiktoken(raw.data(), raw.size(), &vb, &vo, &vi);
        if (!e.empty()) {
            fprintf(stderr, "vocab: %s\n", e.c_str());
            return 2;
        }
        char err[256];
        uint64_t soff = 0;
        void* sim = tks_create(vb.data(), vo.data(), vi.data(), vi.size(), vb.data(), &soff, vi.data(), 0, argv[argc - 1], err, sizeof err);
        if (!sim) {
            fprintf(stderr, "tks_create: %s\n", err);
            return 2;
        }
        std::vector<uint32_t> out(4096);
        for (int r = 0; r < rounds * 50; ++r) {
            std::vector<uint8_t> piece;
            const int parts = 1 + rnd() % 4;
            for (int k = 0; k < parts; ++k) {
                const uint32_t t = rnd() % vi.size();
                uint64_t a = vo[t], b = vo[t + 1];
                if (rnd() % 4 == 0 && b - a > 1) b = a + 1 + rnd() % (b - a - 1);  // a prefix
                piece.insert(piece.end(), vb.begin() + a, vb.begin() + b);
                if (rnd() % 8 == 0) piece.push_back((uint8_t)rnd());
            }
            if (piece.size() > 1000) piece.resize(1000);
            piece.shrink_to_fit();
            tks_lookup(sim, piece.data(), (uint32_t)piece.size());
            tks_encode_piece(sim, piece.data(), (uint32_t)piece.size(), out.data());
            runs += 2;
        }
        tks_destroy(sim);
    }
    printf("ok %llu\n", (unsigned long long)runs);
    return 0;
}
