// pat_str -> TkPat: the split patterns this library has scanners for (reference: the regex compiled once per Encoding,
// src/lib.rs:623; the stock strings are tiktoken_ext/openai_public.py:9-14,89,104-114).
//
// The scanners are hand-written per FAMILY of alternatives -- r50k/gpt2, cl100k, o200k -- and parametrised inside a family by
//   * the contraction list behind the apostrophe (one or two ASCII letters each) and whether it is case-insensitive,
//   * the longest digit group: \p{N}{1,k}, \p{N} (k = 1, e.g. Qwen2's pattern) or \p{N}+,
//   * the suffix set behind a run of "other" chars: [\r\n]*, [\r\n/]*, /* or nothing,
//   * the white-space tail: with or without \s++$ ahead of the newline rule, with or without the newline rule \s*[\r\n]+.
// The parser splits the pattern into its top-level alternatives, ignores possessive markers and the spellings the stock strings
// themselves vary in ([\p{L}] for \p{L}, \s*[\r\n] for \s*[\r\n]+, a trailing \s for \s+, the three ways of writing the
// contraction list), and checks the alternatives of the family one by one.  Anything else is refused with the alternative that
// was not understood.  A pattern whose parameters are the stock ones gets the kernels specialised for that stock pattern.
//
// The kernels cut the text at "certain" piece starts: class pairs (previous char, this char) between which EVERY pattern of the
// family has a boundary whatever stands to the left.  The per-family table was derived for the stock patterns; for any other
// parameter set it is re-derived here: the sequential scanner splits all short strings over class representatives (plus the
// letters of the contraction list) and random longer ones, and every pair that occurs somewhere without a boundary is dropped from
// the family's table (fewer certain starts only mean longer chains for the scanners, never a different split).
#include <string.h>

#include <string>
#include <vector>

#include "tk_device.h"
#include "tk_tables.h"
#include "tk_unicode_tables.inc"

namespace {

// The pattern without possessive markers (a '+' right behind a quantifier: ?, *, +, {m,n}).  A marker is dropped only where it cannot
// change a match: the quantified atom is one of the families' atoms whose class is disjoint from everything that may follow it in a
// member of the family (a give-back never helps there), or it is \s with nothing behind it but `$` or the end of the alternative.
// Anywhere else -- `\s++(?!\S)` matches at the end of the text only, `\s*+[\r\n]+` never, `[\p{Lu}..\p{Lo}]*+[\p{Ll}..\p{Lo}]+` fails
// on a run of \p{Lo} -- *unsafe is set and the caller leaves the pattern to the generic engine, which implements possessive repeats.
std::string strip_possessive(const std::string& s, bool* unsafe) {
    std::string o, atom;
    bool in_class = false, prev_quant = false, prev_open = false;
    *unsafe = false;
    size_t class_from = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '\\' && i + 1 < s.size()) {  // an escape is one token: \s, \p{L}
            size_t j = i + 2;
            if ((s[i + 1] == 'p' || s[i + 1] == 'P') && j < s.size() && s[j] == '{') {
                while (j < s.size() && s[j] != '}') ++j;
                ++j;
            }
            o.append(s, i, j - i);
            if (!in_class) atom.assign(s, i, j - i);
            i = j - 1;
            prev_quant = prev_open = false;
            continue;
        }
        if (in_class) {
            if (c == ']') {
                in_class = false;
                atom.assign(s, class_from, i - class_from + 1);
            }
            o += c;
            prev_quant = prev_open = false;
            continue;
        }
        if (c == '[') {
            in_class = true;
            class_from = i;
            o += c;
            prev_quant = prev_open = false;
            continue;
        }
        if (c == '{') {  // {m,n}
            size_t j = s.find('}', i);
            if (j == std::string::npos) j = s.size() - 1;
            o.append(s, i, j - i + 1);
            i = j;
            prev_quant = true;
            prev_open = false;
            continue;
        }
        if (c == '+' && prev_quant) {  // possessive marker
            prev_quant = false;
            static const char* const SAFE[] = {" ", "[^\\r\\n\\p{L}\\p{N}]", "\\p{L}", "[\\p{L}]", "\\p{N}", "[\\p{N}]", "[^\\s\\p{L}\\p{N}]", "[\\r\\n]", "[\\n\\r]",
                                               "[\\r\\n/]", "[/\\r\\n]", "[\\n\\r/]", "/", "[/]"};
            bool ok = false;
            for (const char* a : SAFE) ok = ok || atom == a;
            if (atom == "\\s") ok = i + 1 >= s.size() || s[i + 1] == '$' || s[i + 1] == '|';
            if (!ok) *unsafe = true;
            continue;
        }
        const bool quant = (c == '?' && !prev_open) || c == '*' || c == '+';
        if (!quant) atom.assign(1, c);
        o += c;
        prev_open = c == '(';
        prev_quant = quant;
    }
    return o;
}

void replace_all(std::string& s, const std::string& a, const std::string& b) {
    for (size_t p = 0; (p = s.find(a, p)) != std::string::npos; p += b.size()) s.replace(p, a.size(), b);
}

std::vector<std::string> split_top(const std::string& s) {
    std::vector<std::string> out;
    std::string cur;
    int par = 0, cls = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '\\' && i + 1 < s.size()) {
            cur += c;
            cur += s[++i];
            continue;
        }
        if (cls) {
            if (c == ']') cls = 0;
        } else if (c == '[') {
            cls = 1;
        } else if (c == '(') {
            ++par;
        } else if (c == ')') {
            --par;
        } else if (c == '|' && par == 0) {
            out.push_back(cur);
            cur.clear();
            continue;
        }
        cur += c;
    }
    out.push_back(cur);
    return out;
}

struct Contr {
    uint32_t c1 = 0;
    std::vector<uint32_t> two;
    bool ok = true;
    std::string why;
    void add(const std::string& letters) {
        for (char ch : letters)
            if (ch < 'a' || ch > 'z') {
                ok = false;
                why = "contraction letters must be lower-case ASCII letters: '" + letters + "'";
                return;
            }
        if (letters.size() == 1) c1 |= 1u << (letters[0] - 'a');
        else if (letters.size() == 2) {
            const uint32_t k = ((uint32_t)letters[0] << 8) | (uint32_t)letters[1];
            for (uint32_t t : two)
                if (t == k) return;
            two.push_back(k);
        } else {
            ok = false;
            why = "contractions of one or two letters only: '" + letters + "'";
        }
    }
};

// items of a contraction list: "[sdmt]", "ll", "'s" (with_apostrophe)
bool parse_contr_items(const std::string& list, bool with_apostrophe, Contr* c) {
    for (const std::string& it0 : split_top(list)) {
        std::string it = it0;
        if (with_apostrophe) {
            if (it.empty() || it[0] != '\'') return false;
            it = it.substr(1);
        }
        if (it.empty()) return false;
        if (it[0] == '[') {
            if (it.back() != ']') return false;
            for (size_t i = 1; i + 1 < it.size(); ++i) c->add(std::string(1, it[i]));
        } else {
            c->add(it);
        }
        if (!c->ok) return false;
    }
    return true;
}

// "'(?:LIST)" / "'(?i:LIST)" / "(?i:'a|'b)" / "(?:'a|'b)"; *optional: the group carries a trailing '?'
bool parse_contr_group(const std::string& alt, Contr* c, bool* ci, bool* optional) {
    std::string a = alt;
    *optional = false;
    if (!a.empty() && a.back() == '?' && a.size() >= 2 && a[a.size() - 2] == ')') {
        *optional = true;
        a.pop_back();
    }
    bool lead_ap = false;
    if (!a.empty() && a[0] == '\'') {
        lead_ap = true;
        a = a.substr(1);
    }
    if (a.size() < 5 || a[0] != '(' || a[1] != '?' || a.back() != ')') return false;
    size_t colon = a.find(':');
    if (colon == std::string::npos) return false;
    const std::string flags = a.substr(2, colon - 2);
    if (flags == "i") *ci = true;
    else if (flags.empty()) *ci = false;
    else return false;
    return parse_contr_items(a.substr(colon + 1, a.size() - colon - 2), !lead_ap, c);
}

struct HostAcc {
    const uint8_t* cls_;
    const uint8_t* text;
    uint64_t n;
    uint32_t cls(uint64_t pos) const { return pos >= n ? (uint32_t)TK_C_END : cls_[pos]; }
    uint32_t byte(uint64_t pos) const { return pos < n ? text[pos] : 0u; }
};

// piece starts of `text` under `pat` by the sequential scanner
void split_seq(const TkTables& T, const TkPat& pat, const std::vector<uint8_t>& text, std::vector<uint8_t>* cls, std::vector<uint8_t>* is_start) {
    const uint64_t n = text.size();
    cls->assign(n + 8, 0);
    is_start->assign(n + 1, 0);
    std::vector<uint8_t> padded(text);
    padded.resize(n + 8, 0);
    for (uint64_t i = 0; i < n; ++i) (*cls)[i] = (uint8_t)tk_classify_text(T, padded.data(), i, n);
    if (n) (*cls)[0] |= TK_F_HARD;
    HostAcc acc{cls->data(), padded.data(), n};
    uint64_t p = 0;
    while (p < n) {
        (*is_start)[p] = 1;
        uint64_t e = tk_piece_end(acc, p, pat);
        if (e <= p) e = tk_next_char(acc, p);
        p = e > n ? n : e;
    }
}

// drops from `cert` every class pair (previous char, this char) that occurs in `text` without a piece boundary between the two
void relax_certain(const TkTables& T, const TkPat& pat, const std::vector<uint8_t>& text, uint16_t* cert) {
    std::vector<uint8_t> cls, st;
    split_seq(T, pat, text, &cls, &st);
    const uint64_t n = text.size();
    uint32_t prev = 16;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t c = cls[i] & 15u;
        if (c == TK_C_CONT) continue;
        if (prev != 16 && !st[i]) cert[prev] &= (uint16_t)~(1u << c);
        prev = c;
    }
}

}  // namespace

void tk_derive_certain(const TkPat& pat, uint16_t* cert) {
    for (uint32_t a = 0; a < 16; ++a) cert[a] = (uint16_t)tk_certain_mask(pat.fam(), a);  // the family's table: pairs are only ever dropped
    TkTables T;
    memset(&T, 0, sizeof T);
    T.uc_stage1 = tk_uc_stage1;
    T.uc_stage2 = tk_uc_stage2;
    // one or more chars per class; the letters of the contraction list in both cases
    std::vector<std::string> sym = {"\n", "\r", " ", "\t", "\xC2\xA0", "X", "x", "\xC7\x85" /* U+01C5, Lt */, "\xCA\xB0" /* U+02B0, Lm */,
                                    "\xCC\x81" /* U+0301, Mn */, "1", "\xC2\xB2" /* superscript two, No */, "'", "/", "!", "\xE2\x82\xAC", "\xC5\xBF" /* long s */};
    uint32_t letters = pat.c1;
    for (uint32_t i = 0; i < pat.n2(); ++i) letters |= (1u << ((pat.two(i) >> 8) - 'a')) | (1u << ((pat.two(i) & 0xFF) - 'a'));
    for (int l = 0; l < 26; ++l)
        if ((letters >> l) & 1u) {
            sym.push_back(std::string(1, (char)('a' + l)));
            sym.push_back(std::string(1, (char)('A' + l)));
        }
    const size_t S = sym.size();
    // strings of four symbols: over the symbols that matter most (one per class that takes part in a rule, contraction letters)
    std::vector<std::string> sym4 = {"\n", " ", "\t", "X", "x", "\xCC\x81", "1", "'", "/", "!"};
    for (size_t k = 17; k < S && sym4.size() < 18; ++k) sym4.push_back(sym[k]);
    std::vector<uint8_t> text;
    for (int len = 1; len <= 4; ++len) {
        const std::vector<std::string>& al = len <= 3 ? sym : sym4;
        const size_t A = al.size();
        size_t total = 1;
        for (int k = 0; k < len; ++k) total *= A;
        for (size_t code = 0; code < total; ++code) {
            text.clear();
            size_t x = code;
            for (int k = 0; k < len; ++k) {
                const std::string& sy = al[x % A];
                text.insert(text.end(), sy.begin(), sy.end());
                x /= A;
            }
            relax_certain(T, pat, text, cert);
        }
    }
    // random longer strings
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < 20000; ++it) {
        text.clear();
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const int len = 5 + (int)((rng >> 33) % 10);
        for (int k = 0; k < len; ++k) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            const std::string& sy = sym[(rng >> 33) % S];
            text.insert(text.end(), sy.begin(), sy.end());
        }
        relax_certain(T, pat, text, cert);
    }
}

std::string tk_parse_pattern(const char* pat_str, TkPat* out, uint16_t* cert_out) {
    if (!pat_str) return "pat_str is null";
    bool possessive_matters = false;
    std::string s = strip_possessive(pat_str, &possessive_matters);
    if (possessive_matters)
        return std::string("unsupported pat_str: a possessive quantifier at a place where it changes the matches (the scanner families "
                           "are written for the backtracking forms); pattern: ") + pat_str;
    replace_all(s, "[\\p{L}]", "\\p{L}");
    replace_all(s, "[\\p{N}]", "\\p{N}");
    std::vector<std::string> alts = split_top(s);
    size_t i = 0;
    auto refuse = [&](const std::string& what) {
        return "unsupported pat_str: " + what + " (supported: the r50k/gpt2, cl100k and o200k patterns and their variations in the "
               "contraction list, the digit group length, the suffix set after punctuation and the white-space rules); pattern: " + pat_str;
    };
    auto have = [&](const char* lit) { return i < alts.size() && alts[i] == lit; };
    static const char* const O2_WORD_A = "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+";
    static const char* const O2_WORD_B = "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*";
    Contr contr;
    bool ci = false;
    int fam = -1;
    // ---- contractions at the front (r50k, cl100k families)
    if (!alts.empty() && alts[0].compare(0, strlen(O2_WORD_A), O2_WORD_A) == 0) {
        fam = TK_PAT_O200K;
        bool have_c[2] = {false, false};
        Contr cc[2];
        bool cis[2] = {false, false};
        const char* const words[2] = {O2_WORD_A, O2_WORD_B};
        for (int w = 0; w < 2; ++w) {
            if (i >= alts.size() || alts[i].compare(0, strlen(words[w]), words[w]) != 0) return refuse("expected the o200k word alternative " + std::string(words[w]));
            const std::string rest = alts[i].substr(strlen(words[w]));
            if (!rest.empty()) {
                bool opt = false;
                if (!parse_contr_group(rest, &cc[w], &cis[w], &opt) || !opt)
                    return refuse(cc[w].ok ? "cannot read the contraction suffix '" + rest + "'" : cc[w].why);
                have_c[w] = true;
            }
            ++i;
        }
        if (have_c[0] != have_c[1] || cc[0].c1 != cc[1].c1 || cc[0].two != cc[1].two || cis[0] != cis[1])
            return refuse("the two word alternatives carry different contraction suffixes");
        contr = cc[0];
        ci = cis[0];
    } else {
        // "'(?i:...)" | "(?i:'s|...)" | bare "'s" alternatives
        bool opt = false;
        if (i < alts.size() && (alts[i].compare(0, 2, "'(") == 0 || alts[i].compare(0, 2, "(?") == 0)) {
            if (!parse_contr_group(alts[i], &contr, &ci, &opt) || opt) return refuse(contr.ok ? "cannot read the contraction alternative '" + alts[i] + "'" : contr.why);
            ++i;
        } else {
            while (i < alts.size() && alts[i].size() >= 2 && alts[i][0] == '\'' && alts[i].find_first_of("\\[(") == std::string::npos) {
                contr.add(alts[i].substr(1));
                if (!contr.ok) return refuse(contr.why);
                ++i;
            }
        }
        if (have(" ?\\p{L}+")) fam = TK_PAT_R50K;
        else if (have("[^\\r\\n\\p{L}\\p{N}]?\\p{L}+")) fam = TK_PAT_CL100K;
        else return refuse(i < alts.size() ? "alternative '" + alts[i] + "' is not the letter alternative of a known family" : "no letter alternative");
        ++i;
    }
    if (contr.two.size() > 4) return refuse("at most four two-letter contractions");
    for (uint32_t t : contr.two) {
        if ((contr.c1 >> ((t >> 8) - 'a')) & 1u) return refuse("a one-letter contraction is the beginning of a two-letter one (the result would depend on their order)");
        if (ci && ((t >> 8) == 's' || (t & 0xFF) == 's' || (t >> 8) == 'k' || (t & 0xFF) == 'k'))
            return refuse("case-insensitive two-letter contractions with 's' or 'k' (U+017F and U+212A fold to them)");
    }
    if (ci && ((contr.c1 >> ('k' - 'a')) & 1u)) return refuse("a case-insensitive contraction 'k' (U+212A folds to it)");
    // ---- digits
    uint32_t digits = 0;
    if (fam == TK_PAT_R50K) {
        if (!have(" ?\\p{N}+")) return refuse("expected ' ?\\p{N}+'");
        ++i;
    } else {
        if (i >= alts.size()) return refuse("no digit alternative");
        const std::string& d = alts[i];
        if (d == "\\p{N}+") digits = 0;
        else if (d == "\\p{N}") digits = 1;
        else if (d.compare(0, 8, "\\p{N}{1,") == 0 && d.back() == '}') {
            const std::string k = d.substr(8, d.size() - 9);
            if (k.empty() || k.size() > 3 || k.find_first_not_of("0123456789") != std::string::npos) return refuse("cannot read the digit group '" + d + "'");
            digits = (uint32_t)atoi(k.c_str());
            if (digits < 1 || digits > 255) return refuse("digit groups of 1..255 digits");
        } else {
            return refuse("alternative '" + d + "' is not a digit group (\\p{N}{1,k}, \\p{N} or \\p{N}+)");
        }
        ++i;
    }
    // ---- other chars (+ suffix set)
    uint32_t suffix = 0;
    {
        static const char* const OTHER = " ?[^\\s\\p{L}\\p{N}]+";
        if (i >= alts.size() || alts[i].compare(0, strlen(OTHER), OTHER) != 0) return refuse("expected ' ?[^\\s\\p{L}\\p{N}]+'");
        const std::string rest = alts[i].substr(strlen(OTHER));
        if (rest.empty()) suffix = 0;
        else if (fam == TK_PAT_R50K) return refuse("a suffix set in the r50k family");
        else if (rest == "[\\r\\n]*" || rest == "[\\n\\r]*") suffix = 1;
        else if (rest == "[\\r\\n/]*" || rest == "[/\\r\\n]*" || rest == "[\\n\\r/]*") suffix = 3;
        else if (rest == "/*" || rest == "[/]*") suffix = 2;
        else return refuse("suffix set '" + rest + "' (supported: [\\r\\n]*, [\\r\\n/]*, /*)");
        ++i;
    }
    // ---- white space
    bool dollar = false, nl = false;
    if (have("\\s+$")) {
        dollar = true;
        ++i;
    }
    if (have("\\s*[\\r\\n]+") || have("\\s*[\\r\\n]")) {
        if (fam == TK_PAT_R50K) return refuse("a newline rule in the r50k family");
        nl = true;
        ++i;
    }
    if (!have("\\s+(?!\\S)")) return refuse(i < alts.size() ? "alternative '" + alts[i] + "' where '\\s+(?!\\S)' was expected" : "'\\s+(?!\\S)' is missing");
    ++i;
    if (!(have("\\s+") || have("\\s"))) return refuse("the last alternative must be '\\s+' or '\\s'");
    ++i;
    if (i != alts.size()) return refuse("unexpected alternative '" + alts[i] + "'");
    if (fam == TK_PAT_R50K) dollar = true;  // (without a newline rule \s+(?!\S) takes the whole trailing run anyway)
    if (!nl) dollar = true;
    TkPat p;
    p.w0 = (uint32_t)fam | (ci ? 4u : 0u) | (suffix << 3) | (dollar ? 64u : 0u) | (nl ? 128u : 0u) | (digits << 8) | ((uint32_t)contr.two.size() << 16);
    p.c1 = contr.c1;
    p.c2[0] = p.c2[1] = 0;
    // two-letter contractions in a canonical order (so that equal lists compare equal)
    std::vector<uint32_t> two = contr.two;
    for (size_t a = 0; a < two.size(); ++a)
        for (size_t b = a + 1; b < two.size(); ++b)
            if (two[b] < two[a]) std::swap(two[a], two[b]);
    for (size_t k = 0; k < two.size(); ++k) p.c2[k >> 1] |= two[k] << (16u * (k & 1u));
    // the stock pattern of the family (same canonical order)
    TkPat stock = tk_stock_pat(fam);
    {
        uint32_t st[3] = {stock.two(0), stock.two(1), stock.two(2)};
        for (int a = 0; a < 3; ++a)
            for (int b = a + 1; b < 3; ++b)
                if (st[b] < st[a]) std::swap(st[a], st[b]);
        stock.c2[0] = st[0] | (st[1] << 16);
        stock.c2[1] = st[2];
    }
    uint16_t cert[16];
    if (!p.same_as(stock)) {
        p.w0 |= 32u;  // generic kernels
        tk_derive_certain(p, cert);
    } else {
        p = stock;
        for (uint32_t a = 0; a < 16; ++a) cert[a] = (uint16_t)tk_certain_mask(fam, a);
    }
    if (cert_out) memcpy(cert_out, cert, sizeof cert);
    *out = p;
    return "";
}
