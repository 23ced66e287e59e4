// Deterministic synthetic corpus generator (bench/test support, not part of the encode path).
//
// The reference's benchmark protocol (scripts/benchmark.py:15-26) times
// encode_ordinary_batch over a list of documents; no corpus ships with it and this
// environment has no network, so SURVEY.md §8(d) defines seeded synthetic corpora instead:
// log-normal document lengths (median 2 KiB, sigma 1.0, clipped to [64 B, 256 KiB]) and a
// byte mix of Latin prose, code, Cyrillic, Greek/Arabic/Hebrew/Devanagari/Thai, CJK, emoji /
// symbols / combining marks, numbers / URLs / whitespace runs.  Only code points assigned in
// Unicode <= 13 are emitted so that every regex engine agrees on their classes.
//
// C ABI:
//   tkc_generate(seed, mix, total_bytes, out, doc_off, max_docs, &n_docs, n_threads)
// Documents are packed back to back in `out` (exactly total_bytes bytes); doc_off has
// n_docs+1 entries.  Every document is valid UTF-8.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {  // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    bool chance(double p) { return unit() < p; }
};

void put_cp(std::string& o, uint32_t cp) {
    if (cp < 0x80) {
        o.push_back((char)cp);
    } else if (cp < 0x800) {
        o.push_back((char)(0xC0 | (cp >> 6)));
        o.push_back((char)(0x80 | (cp & 63)));
    } else if (cp < 0x10000) {
        o.push_back((char)(0xE0 | (cp >> 12)));
        o.push_back((char)(0x80 | ((cp >> 6) & 63)));
        o.push_back((char)(0x80 | (cp & 63)));
    } else {
        o.push_back((char)(0xF0 | (cp >> 18)));
        o.push_back((char)(0x80 | ((cp >> 12) & 63)));
        o.push_back((char)(0x80 | ((cp >> 6) & 63)));
        o.push_back((char)(0x80 | (cp & 63)));
    }
}

// A lexicon: word strings + a Zipf sampler over them.
struct Lexicon {
    std::vector<std::string> words;
    std::vector<double> cdf;
    void finish(double s_exp) {
        cdf.resize(words.size());
        double acc = 0;
        for (size_t i = 0; i < words.size(); ++i) {
            acc += 1.0 / std::pow((double)(i + 1), s_exp);
            cdf[i] = acc;
        }
        for (auto& c : cdf) c /= acc;
    }
    const std::string& sample(Rng& r) const {
        double u = r.unit();
        size_t i = std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
        if (i >= words.size()) i = words.size() - 1;
        return words[i];
    }
};

// Build words as 1..max_syl syllables drawn from an onset/nucleus/coda inventory.
Lexicon make_syllabic(uint64_t seed, size_t n_words, const std::vector<std::string>& onset,
                      const std::vector<std::string>& nucleus, const std::vector<std::string>& coda,
                      int max_syl) {
    Rng r(seed);
    Lexicon lx;
    lx.words.reserve(n_words);
    // the most frequent words are short
    for (size_t i = 0; i < n_words; ++i) {
        int lim = i < 50 ? 1 : (i < 2000 ? 2 : max_syl);
        int ns = 1 + (int)r.below((uint32_t)lim);
        std::string w;
        for (int k = 0; k < ns; ++k) {
            if (r.chance(0.8)) w += onset[r.below((uint32_t)onset.size())];
            w += nucleus[r.below((uint32_t)nucleus.size())];
            if (r.chance(0.45)) w += coda[r.below((uint32_t)coda.size())];
        }
        lx.words.push_back(std::move(w));
    }
    lx.finish(1.08);
    return lx;
}

std::vector<std::string> cps(std::initializer_list<std::pair<uint32_t, uint32_t>> ranges) {
    std::vector<std::string> v;
    for (auto& pr : ranges)
        for (uint32_t c = pr.first; c <= pr.second; ++c) {
            std::string s;
            put_cp(s, c);
            v.push_back(s);
        }
    return v;
}

Lexicon make_from_alphabet(uint64_t seed, size_t n_words, const std::vector<std::string>& letters,
                           int min_len, int max_len, const std::vector<std::string>* marks = nullptr,
                           double mark_p = 0.0) {
    Rng r(seed);
    // letter frequencies are themselves Zipfian so BPE finds structure
    std::vector<double> lcdf(letters.size());
    double acc = 0;
    for (size_t i = 0; i < letters.size(); ++i) {
        acc += 1.0 / std::pow((double)(i + 1), 0.9);
        lcdf[i] = acc;
    }
    for (auto& c : lcdf) c /= acc;
    Lexicon lx;
    for (size_t i = 0; i < n_words; ++i) {
        int hi = i < 100 ? std::max(min_len, std::min(max_len, 2)) : max_len;
        int len = min_len + (int)r.below((uint32_t)(hi - min_len + 1));
        std::string w;
        for (int k = 0; k < len; ++k) {
            size_t j = std::lower_bound(lcdf.begin(), lcdf.end(), r.unit()) - lcdf.begin();
            if (j >= letters.size()) j = letters.size() - 1;
            w += letters[j];
            if (marks && r.chance(mark_p)) w += (*marks)[r.below((uint32_t)marks->size())];
        }
        lx.words.push_back(std::move(w));
    }
    lx.finish(1.05);
    return lx;
}

struct World {
    Lexicon latin, cyr, greek, arabic, hebrew, deva, thai, han, kana, hangul, ident;
    std::vector<std::string> emoji, symbols, comb, tlds;
    World() {
        std::vector<std::string> on = {"b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "qu", "r", "s", "t",
                                       "v", "w", "x", "y", "z", "bl", "br", "ch", "cl", "cr", "dr", "fl", "fr", "gl",
                                       "gr", "pl", "pr", "sc", "sh", "sk", "sl", "sm", "sn", "sp", "st", "str", "sw",
                                       "th", "tr", "tw", "wh", "wr", "ph", "kn"};
        std::vector<std::string> nu = {"a", "e", "i", "o", "u", "ai", "au", "ea", "ee", "ei", "ie", "io", "oa", "oo",
                                       "ou", "ue", "y", "\xC3\xA9", "\xC3\xA4", "\xC3\xB6", "\xC3\xBC", "\xC3\xB1"};
        std::vector<std::string> co = {"b", "d", "g", "k", "l", "m", "n", "p", "r", "s", "t", "x", "ck", "ct", "ft",
                                       "ld", "ll", "lt", "mp", "nd", "ng", "nk", "nt", "pt", "rd", "rk", "rm", "rn",
                                       "rs", "rt", "sh", "sk", "sm", "sp", "ss", "st", "th", "tion", "ing", "ed", "er",
                                       "ly", "ment", "ness", "ous", "ive", "able", "al", "ic"};
        latin = make_syllabic(0xA11CE, 600000, on, nu, co, 5);
        ident = make_syllabic(0x1DE27, 40000, on, nu, co, 3);
        cyr = make_from_alphabet(0xC1, 60000, cps({{0x430, 0x44F}}), 1, 11);
        greek = make_from_alphabet(0xC2, 30000, cps({{0x3B1, 0x3C1}, {0x3C3, 0x3C9}}), 1, 10);
        auto harakat = cps({{0x64B, 0x652}});
        arabic = make_from_alphabet(0xC3, 30000, cps({{0x621, 0x63A}, {0x641, 0x64A}}), 2, 8, &harakat, 0.08);
        hebrew = make_from_alphabet(0xC4, 20000, cps({{0x5D0, 0x5EA}}), 2, 7);
        auto matras = cps({{0x93E, 0x94D}});
        deva = make_from_alphabet(0xC5, 30000, cps({{0x915, 0x939}, {0x905, 0x914}}), 1, 6, &matras, 0.55);
        auto thai_marks = cps({{0xE31, 0xE31}, {0xE34, 0xE3A}, {0xE47, 0xE4E}});
        thai = make_from_alphabet(0xC6, 30000, cps({{0xE01, 0xE2E}, {0xE30, 0xE30}, {0xE32, 0xE33}, {0xE40, 0xE44}}),
                                  2, 7, &thai_marks, 0.3);
        // ~6k Han characters, Zipfian; words are 1-3 characters
        han = make_from_alphabet(0xC7, 80000, cps({{0x4E00, 0x65FF}}), 1, 3);
        kana = make_from_alphabet(0xC8, 30000, cps({{0x3041, 0x3093}, {0x30A1, 0x30F6}}), 1, 5);
        hangul = make_from_alphabet(0xC9, 40000, cps({{0xAC00, 0xB7FF}}), 1, 4);
        emoji = cps({{0x1F600, 0x1F64F}, {0x1F300, 0x1F320}, {0x1F680, 0x1F6A0}, {0x2600, 0x2615}});
        symbols = cps({{0xA9, 0xA9}, {0xAE, 0xAE}, {0x2122, 0x2122}, {0x2190, 0x2194}, {0x2264, 0x2265},
                       {0xB1, 0xB1}, {0xD7, 0xD7}, {0xF7, 0xF7}, {0x2022, 0x2022}, {0x2026, 0x2026},
                       {0x2013, 0x2014}, {0x201C, 0x201D}, {0x2018, 0x2019}, {0xAB, 0xAB}, {0xBB, 0xBB},
                       {0x20AC, 0x20AC}, {0xA3, 0xA3}, {0xA5, 0xA5}, {0xB0, 0xB0}, {0xA7, 0xA7}, {0xB6, 0xB6}});
        comb = cps({{0x300, 0x304}, {0x308, 0x308}, {0x30A, 0x30A}, {0x327, 0x327}});
        tlds = {"com", "org", "net", "io", "dev", "edu", "gov", "co.uk", "de", "fr", "jp"};
    }
};

const World& world() {
    static World w;
    return w;
}

enum Kind { K_LATIN, K_CODE, K_CYR, K_GREEK, K_ARABIC, K_HEBREW, K_DEVA, K_THAI, K_HAN, K_KANA, K_HANGUL,
            K_EMOJI, K_NUMWS, K_MARKUP, K_COUNT };

std::string capitalised(const std::string& w) {
    std::string o = w;
    if (!o.empty() && o[0] >= 'a' && o[0] <= 'z') o[0] = (char)(o[0] - 32);
    return o;
}
std::string upper(const std::string& w) {
    std::string o = w;
    for (auto& c : o)
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    return o;
}

void number(Rng& r, std::string& o) {
    switch (r.below(12)) {
        case 0: o += std::to_string(1900 + r.below(150)); break;
        case 1: o += std::to_string(r.below(100)); break;
        case 2: o += std::to_string(r.below(1000000)); break;
        case 3: o += std::to_string(r.below(1000)) + "." + std::to_string(r.below(100)); break;
        case 4: o += std::to_string(2000 + r.below(30)) + "-" + (r.chance(.5) ? "0" : "1") + std::to_string(r.below(3)) + "-" + std::to_string(10 + r.below(19)); break;
        case 5: o += std::to_string(r.below(24)) + ":" + std::to_string(10 + r.below(50)); break;
        case 6: o += "$" + std::to_string(r.below(999)) + "," + std::to_string(100 + r.below(900)) + "." + std::to_string(10 + r.below(90)); break;
        case 7: o += std::to_string(r.below(101)) + "%"; break;
        case 8: { char b[24]; snprintf(b, sizeof b, "0x%X", r.below(1u << 24)); o += b; break; }
        case 9: o += std::to_string(r.next() % 100000000000ull); break;
        case 10: { int n = 1 + r.below(4); for (int i = 0; i < n; ++i) put_cp(o, 0x660 + r.below(10)); break; }
        default: put_cp(o, r.chance(.5) ? 0xB2 + r.below(2) : (r.chance(.5) ? 0xBD : 0x2163)); break;
    }
}

void url(Rng& r, std::string& o) {
    const World& w = world();
    o += r.chance(.8) ? "https://" : "http://";
    if (r.chance(.6)) o += "www.";
    o += w.ident.sample(r);
    o += ".";
    o += w.tlds[r.below((uint32_t)w.tlds.size())];
    int segs = r.below(4);
    for (int i = 0; i < segs; ++i) {
        o += "/";
        o += w.ident.sample(r);
        if (r.chance(.2)) { o += "-"; o += w.ident.sample(r); }
    }
    if (r.chance(.3)) { o += "?"; o += w.ident.sample(r); o += "="; o += std::to_string(r.below(1000)); }
    if (r.chance(.1)) { o += "&"; o += w.ident.sample(r); o += "="; o += w.ident.sample(r); }
}

void latin_sentence(Rng& r, std::string& o) {
    const World& w = world();
    static const char* contr[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
    int n = 3 + r.below(22);
    for (int i = 0; i < n; ++i) {
        const std::string& word = w.latin.sample(r);
        double u = r.unit();
        if (i == 0 || u < 0.06) o += capitalised(word);
        else if (u < 0.075) o += upper(word);
        else if (u < 0.085) { o += capitalised(word); o += capitalised(w.latin.sample(r)); }  // CamelCase
        else o += word;
        if (r.chance(0.035)) {
            const char* c = contr[r.below(7)];
            if (r.chance(0.1)) { std::string t = c; o += upper(t); }
            else if (r.chance(0.05)) { o += "\xE2\x80\x99"; o += (c + 1); }  // typographic apostrophe
            else o += c;
        }
        if (r.chance(0.01)) o += w.comb[r.below((uint32_t)w.comb.size())];
        if (i + 1 < n) {
            double v = r.unit();
            if (v < 0.08) o += ", ";
            else if (v < 0.09) o += "; ";
            else if (v < 0.10) o += ": ";
            else if (v < 0.11) o += " - ";
            else if (v < 0.12) { o += " ("; }
            else if (v < 0.13) { o += ") "; }
            else if (v < 0.14) { o += " \""; }
            else if (v < 0.15) { o += "\" "; }
            else if (v < 0.16) { o += " "; number(r, o); o += " "; }
            else if (v < 0.165) { o += " "; url(r, o); o += " "; }
            else if (v < 0.17) { o += "  "; }
            else if (v < 0.172) { o += "\xC2\xA0"; }
            else if (v < 0.18) { o += "-"; }
            else if (v < 0.185) { o += "/"; }
            else if (v < 0.19) { o += " "; o += w.symbols[r.below((uint32_t)w.symbols.size())]; o += " "; }
            else o += " ";
        }
    }
    double v = r.unit();
    o += v < 0.8 ? "." : (v < 0.88 ? "?" : (v < 0.94 ? "!" : (v < 0.97 ? "..." : ":")));
}

void paragraph_break(Rng& r, std::string& o) {
    double v = r.unit();
    if (v < 0.55) o += "\n\n";
    else if (v < 0.75) o += "\n";
    else if (v < 0.82) o += "\r\n\r\n";
    else if (v < 0.86) o += " \n";
    else if (v < 0.90) o += "\n\n\n";
    else if (v < 0.94) o += "\n    ";
    else if (v < 0.97) o += "\n\t";
    else o += " \n \n";
}

void space_script_sentence(Rng& r, std::string& o, const Lexicon& lx, const char* stop, bool caps_cyr) {
    int n = 3 + r.below(18);
    for (int i = 0; i < n; ++i) {
        const std::string& word = lx.sample(r);
        if (caps_cyr && (i == 0 || r.chance(0.05))) {
            // upper-case the first Cyrillic/Greek letter (2-byte sequences: U+0430.. -> U+0410.., U+03B1.. -> U+0391..)
            unsigned char b0 = (unsigned char)word[0], b1 = (unsigned char)word[1];
            uint32_t cp = ((b0 & 0x1F) << 6) | (b1 & 0x3F);
            if (cp >= 0x430 && cp <= 0x44F) cp -= 0x20;
            else if (cp >= 0x3B1 && cp <= 0x3C9 && cp != 0x3C2) cp -= 0x20;
            put_cp(o, cp);
            o += word.substr(2);
        } else {
            o += word;
        }
        if (i + 1 < n) {
            double v = r.unit();
            if (v < 0.08) o += ", ";
            else if (v < 0.10) { o += " "; number(r, o); o += " "; }
            else o += " ";
        }
    }
    o += stop;
}

void nospace_sentence(Rng& r, std::string& o, const Lexicon& a, const Lexicon* b, const char* comma, const char* stop) {
    int n = 4 + r.below(30);
    for (int i = 0; i < n; ++i) {
        o += (b && r.chance(0.35)) ? b->sample(r) : a.sample(r);
        double v = r.unit();
        if (v < 0.07) o += comma;
        else if (v < 0.085) number(r, o);
        else if (v < 0.095) { o += world().latin.sample(r); }
        else if (v < 0.10) { o += "\xE3\x80\x80"; }  // U+3000
    }
    o += stop;
}

void code_line(Rng& r, std::string& o, int& indent) {
    const World& w = world();
    auto id = [&](std::string& s) {
        const std::string& a = w.ident.sample(r);
        double v = r.unit();
        if (v < 0.3) { s += a; s += "_"; s += w.ident.sample(r); }
        else if (v < 0.5) { s += a; s += capitalised(w.ident.sample(r)); }
        else if (v < 0.55) s += upper(a);
        else s += a;
    };
    if (r.chance(0.15) && indent > 0) indent--;
    if (r.chance(0.1)) { for (int i = 0; i < indent; ++i) o += "\t"; }
    else { for (int i = 0; i < indent * 4; ++i) o += ' '; }
    switch (r.below(10)) {
        case 0: o += "def "; id(o); o += "("; id(o); o += ", "; id(o); o += "=None):"; indent++; break;
        case 1: o += "if ("; id(o); o += (r.chance(.5) ? " == " : " != "); number(r, o); o += ") {"; indent++; break;
        case 2: o += "return "; id(o); o += (r.chance(.5) ? " + " : " * "); id(o); o += ";"; break;
        case 3: o += "// "; latin_sentence(r, o); break;
        case 4: id(o); o += " = "; id(o); o += "["; number(r, o); o += "]"; o += (r.chance(.5) ? ";" : ""); break;
        case 5: o += "for (int i = 0; i < "; id(o); o += "; ++i) {"; indent++; break;
        case 6: o += "}"; if (indent > 0) indent--; break;
        case 7: id(o); o += "->"; id(o); o += "("; o += "\""; o += w.ident.sample(r); o += "\""; o += ");"; break;
        case 8: o += "/* "; id(o); o += " */ "; id(o); o += "::"; id(o); o += "<"; id(o); o += ">();"; break;
        default: o += "# "; for (int i = 0; i < 20 + (int)r.below(60); ++i) o += (r.chance(.5) ? '=' : '-'); break;
    }
    if (indent > 6) indent = 6;
    o += r.chance(0.08) ? "\r\n" : "\n";
    if (r.chance(0.08)) o += "\n";
}

void markup_line(Rng& r, std::string& o) {
    const World& w = world();
    switch (r.below(7)) {
        case 0: o += "<div class=\""; o += w.ident.sample(r); o += "\">"; latin_sentence(r, o); o += "</div>\n"; break;
        case 1: o += "<p>"; latin_sentence(r, o); o += "</p>\n"; break;
        case 2: o += "<a href=\""; url(r, o); o += "\">"; o += w.latin.sample(r); o += "</a>&nbsp;&amp; "; break;
        case 3: o += "# "; o += capitalised(w.latin.sample(r)); o += " "; o += capitalised(w.latin.sample(r)); o += "\n\n"; break;
        case 4: o += "* **"; o += w.latin.sample(r); o += "**: "; latin_sentence(r, o); o += "\n"; break;
        case 5: o += "["; o += w.latin.sample(r); o += "]("; url(r, o); o += ")\n"; break;
        default: o += "| "; o += w.latin.sample(r); o += " | "; number(r, o); o += " | "; number(r, o); o += " |\n"; break;
    }
}

void emit_kind(Rng& r, std::string& o, int kind, int& indent) {
    const World& w = world();
    switch (kind) {
        case K_LATIN: latin_sentence(r, o); o += r.chance(0.2) ? "" : " "; if (r.chance(0.18)) paragraph_break(r, o); break;
        case K_CODE: code_line(r, o, indent); break;
        case K_CYR: space_script_sentence(r, o, w.cyr, ". ", true); if (r.chance(0.15)) paragraph_break(r, o); break;
        case K_GREEK: space_script_sentence(r, o, w.greek, ". ", true); break;
        case K_ARABIC: space_script_sentence(r, o, w.arabic, "\xD8\x9F ", false); break;   // U+061F
        case K_HEBREW: space_script_sentence(r, o, w.hebrew, ". ", false); break;
        case K_DEVA: space_script_sentence(r, o, w.deva, "\xE0\xA5\xA4 ", false); break;    // U+0964 danda
        case K_THAI: nospace_sentence(r, o, w.thai, nullptr, " ", " "); break;
        case K_HAN: nospace_sentence(r, o, w.han, nullptr, "\xEF\xBC\x8C", "\xE3\x80\x82"); if (r.chance(0.1)) o += "\n"; break;
        case K_KANA: nospace_sentence(r, o, w.kana, &w.han, "\xE3\x80\x81", "\xE3\x80\x82"); break;
        case K_HANGUL: space_script_sentence(r, o, w.hangul, ". ", false); break;
        case K_EMOJI: {
            int n = 1 + r.below(6);
            for (int i = 0; i < n; ++i) {
                double v = r.unit();
                if (v < 0.5) o += w.emoji[r.below((uint32_t)w.emoji.size())];
                else if (v < 0.8) { o += w.symbols[r.below((uint32_t)w.symbols.size())]; }
                else { o += w.latin.sample(r); o += w.comb[r.below((uint32_t)w.comb.size())]; }
                if (r.chance(0.5)) o += " ";
            }
            break;
        }
        case K_NUMWS: {
            double v = r.unit();
            if (v < 0.3) { number(r, o); o += r.chance(.5) ? " " : "\t"; }
            else if (v < 0.5) { url(r, o); o += "\n"; }
            else if (v < 0.6) { int n = 2 + r.below(12); for (int i = 0; i < n; ++i) o += ' '; }
            else if (v < 0.7) o += "\r\n";
            else if (v < 0.78) o += "\t\t";
            else if (v < 0.84) o += "\xC2\xA0";
            else if (v < 0.88) o += "\xE3\x80\x80";
            else if (v < 0.93) { o += "\n\n\n"; }
            else { number(r, o); o += ","; number(r, o); o += ","; number(r, o); o += "\n"; }
            break;
        }
        default: markup_line(r, o); break;
    }
}

// Mixes: cumulative byte-share targets per kind.  mix 0 = "mixed UTF-8" (config C2),
// mix 1 = "web text" (configs C3/C4/C5), mix 2 = ASCII prose only.
const double MIXES[3][K_COUNT] = {
    // LATIN CODE  CYR  GREEK ARAB  HEBR  DEVA  THAI  HAN   KANA  HANGUL EMOJI NUMWS MARKUP
    {0.55, 0.10, 0.10, 0.01, 0.012, 0.008, 0.01, 0.01, 0.07, 0.025, 0.025, 0.03, 0.05, 0.00},
    {0.66, 0.05, 0.06, 0.008, 0.01, 0.006, 0.008, 0.008, 0.05, 0.02, 0.02, 0.01, 0.03, 0.09},
    {1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
};

int pick_kind(Rng& r, int mix) {
    double u = r.unit(), acc = 0;
    for (int k = 0; k < K_COUNT; ++k) {
        acc += MIXES[mix][k];
        if (u < acc) return k;
    }
    return K_LATIN;
}

void fill_doc(uint64_t seed, uint64_t doc_index, int mix, uint8_t* dst, uint64_t len) {
    Rng r(seed * 0x100000001B3ull + doc_index * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull);
    std::string o;
    o.reserve((size_t)len + 512);
    // a document has a dominant kind (70 % of its segments) plus a sprinkle from the mix
    int dominant = pick_kind(r, mix);
    int indent = 0;
    while (o.size() < len) {
        int k = r.chance(0.7) ? dominant : pick_kind(r, mix);
        emit_kind(r, o, k, indent);
    }
    // truncate to a char boundary <= len and pad with '.' so the document is exactly len bytes
    size_t cut = (size_t)len;
    while (cut > 0 && ((unsigned char)o[cut] & 0xC0) == 0x80) --cut;
    memcpy(dst, o.data(), cut);
    for (size_t i = cut; i < len; ++i) dst[i] = '.';
}

}  // namespace

extern "C" {

// Returns 0 on success, -1 if max_docs is too small.
int tkc_generate(uint64_t seed, int mix, uint64_t total_bytes, uint8_t* out, uint64_t* doc_off,
                 uint64_t max_docs, uint64_t* n_docs_out, int n_threads) {
    if (mix < 0 || mix > 2) return -2;
    Rng r(seed ^ 0xD0C5EEDull);
    std::vector<uint64_t> offs;
    offs.push_back(0);
    uint64_t pos = 0;
    while (pos < total_bytes) {
        // log-normal: median 2 KiB, sigma 1.0 (Box-Muller)
        double u1 = r.unit(), u2 = r.unit();
        if (u1 < 1e-300) u1 = 1e-300;
        double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
        double l = 2048.0 * std::exp(z);
        uint64_t len = (uint64_t)l;
        if (len < 64) len = 64;
        if (len > 262144) len = 262144;
        if (pos + len > total_bytes) len = total_bytes - pos;
        pos += len;
        offs.push_back(pos);
    }
    uint64_t n_docs = offs.size() - 1;
    if (n_docs > max_docs) return -1;
    for (uint64_t i = 0; i <= n_docs; ++i) doc_off[i] = offs[i];
    *n_docs_out = n_docs;
    (void)world();  // build lexicons once, before the threads start
    if (n_threads < 1) n_threads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            for (uint64_t d = (uint64_t)t; d < n_docs; d += (uint64_t)n_threads)
                fill_doc(seed, d, mix, out + offs[d], offs[d + 1] - offs[d]);
        });
    }
    for (auto& x : th) x.join();
    return 0;
}

}  // extern "C"
