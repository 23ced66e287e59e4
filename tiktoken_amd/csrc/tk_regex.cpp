// pat_str -> program of the generic engine (tk_regex.h).  Reference: Regex::new(pattern) of fancy-regex, src/lib.rs:623 -- a pattern
// outside the three scanner families of tk_pattern.cpp is no longer refused: it is parsed here (the syntax fancy-regex and the Rust
// `regex` crate share with Python `regex`), compiled to a backtracking program and run on the GPU.
//
// Supported: literals, `.`, classes [...] with ranges / escapes / negation / intersection [A&&[^B]] / difference [A--B], \d \s \w \D \S \W, \p{..} \P{..} for General_Category values,
// scripts (\p{Han}, \p{Script=Greek}) and the binary properties of the UCD (\p{Alphabetic}, \p{Emoji}, ...: tk_regex_binprops.inc), the POSIX classes
// [[:alpha:]] [[:^digit:]] (ASCII, as in the Rust `regex` crate),
// the escapes \n \r \t \f \v \xHH \x{H..} \uHHHH \u{H..} \UHHHHHHHH, alternation, groups (capturing ones are plain groups: a split pattern
// has no use for captures), (?: ) (?i: ) (?s: ) (?m: ) (?x: ) (?i) (?s) (?m) (?x) (?-i), atomic groups (?> ), look-ahead (?= ) (?! ), the quantifiers ? * + {m} {m,}
// {m,n} in their greedy, lazy (?) and possessive (+) forms, ^ \A $ \z, \b \B, look-behind of fixed length (?<=ab|c) (?<!\S).  Refused, with the reason:
// look-behind of variable length, back-references,
// the class set operation ~~ and set operations nested in operands, script extensions, case-insensitive matching of
// non-ASCII cased letters, a pattern (or a repeated group) that can match the empty string.
#include "tk_regex.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "tk_regex_host.h"
#include "tk_regex_props.inc"
#include "tk_regex_scripts.inc"
#include "tk_regex_binprops.inc"
#include "tk_regex_casefold.inc"

namespace {

const char* const GC_NAMES[30] = {"Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Mc", "Me", "Nd", "Nl", "No", "Pc", "Pd", "Ps", "Pe",
                                  "Pi", "Pf", "Po", "Sm", "Sc", "Sk", "So", "Zs", "Zl", "Zp", "Cc", "Cf", "Cs", "Co", "Cn"};

uint32_t prop_of(uint32_t cp) {
    if (cp > 0x10FFFFu) cp = 0xFFFDu;
    return tk_rx_stage2[(uint32_t)tk_rx_stage1[cp >> 8] * 256u + (cp & 255u)];
}

struct CharSet {
    bool neg = false;
    uint32_t gcmask = 0, flags = 0;  // flags: 0x20 \s, 0x40 \w
    bool comp = false;               // one complemented term: \S \D \W \P{..} inside a class
    uint32_t cgcmask = 0, cflags = 0;
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    int and_set = -1;  // [A&&B], [A--B]: the set the members also have to be in (an operand of its own in Parser::sets; chains on)
    bool raw(uint32_t cp) const {  // membership before negation and intersection
        const uint32_t pr = prop_of(cp);
        bool in = ((gcmask >> (pr & 31u)) & 1u) || (pr & flags & 0x60u);
        if (!in && comp) in = !(((cgcmask >> (pr & 31u)) & 1u) || (pr & cflags & 0x60u));
        for (size_t i = 0; i < ranges.size() && !in; ++i) in = cp >= ranges[i].first && cp <= ranges[i].second;
        return in;
    }
};

struct Node {
    enum Kind { EMPTY, SET, CAT, ALT, REPEAT, ATOMIC, LOOK, START, END, WORDB, BEHIND } kind = EMPTY;
    int set = -1;
    std::vector<int> kids;
    uint32_t mn = 0, mx = 0;
    int mode = TK_RX_GREEDY;
    bool neg = false;
};

struct Flags {
    bool ci = false, dotall = false, multiline = false, verbose = false;
};

struct Parser {
    std::vector<uint32_t> s;  // the pattern as code points
    size_t i = 0;
    std::vector<Node> nodes;
    std::vector<CharSet> sets;
    std::string err;

    bool fail(const std::string& m) {
        if (err.empty()) err = m + " (at offset " + std::to_string(i) + " of the pattern)";
        return false;
    }
    bool more() const { return i < s.size(); }
    uint32_t peek(size_t k = 0) const { return i + k < s.size() ? s[i + k] : 0u; }
    int add(const Node& n) {
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }
    int add_set(const CharSet& c) {
        sets.push_back(c);
        Node n;
        n.kind = Node::SET;
        n.set = (int)sets.size() - 1;
        return add(n);
    }

    // ---- sets
    bool member(const CharSet& c, uint32_t cp) const {
        bool in = c.raw(cp);
        for (int k = c.and_set; in && k >= 0; k = sets[k].and_set) in = sets[k].raw(cp) != sets[k].neg;
        return in != c.neg;
    }
    // simple case folding (tk_regex_casefold.inc: what Python `regex` matches under (?i), one char for one char, minus the Turkic i's -- the Rust
    // crate's CaseFolding C + S): the equivalents of code points [lo, hi] join the set
    static void add_equivalents(CharSet& c, uint32_t lo, uint32_t hi) {
        uint32_t a = 0, b = TK_RX_NCASEFOLD;
        while (a < b) {  // first entry with code point >= lo
            const uint32_t m = (a + b) / 2;
            if (tk_rx_casefold[m][0] < lo) a = m + 1;
            else b = m;
        }
        for (; a < TK_RX_NCASEFOLD && tk_rx_casefold[a][0] <= hi; ++a) c.ranges.push_back({tk_rx_casefold[a][1], tk_rx_casefold[a][1]});
    }
    bool add_char(CharSet& c, uint32_t cp, bool ci) {
        c.ranges.push_back({cp, cp});
        if (!ci) return true;
        if (cp < 128u && ((cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z'))) c.ranges.push_back({cp ^ 0x20u, cp ^ 0x20u});
        add_equivalents(c, cp, cp);  // (beyond ASCII, and the two letters ASCII shares a class with: U+017F long s, U+212A Kelvin sign)
        return true;
    }
    bool add_range(CharSet& c, uint32_t lo, uint32_t hi, bool ci) {
        if (lo > hi) return fail("class range out of order");
        if (hi > 0x10FFFFu) return fail("class range beyond U+10FFFF");
        c.ranges.push_back({lo, hi});
        if (!ci) return true;
        for (uint32_t cp = lo; cp <= hi && cp < 128u; ++cp)
            if ((cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z')) c.ranges.push_back({cp ^ 0x20u, cp ^ 0x20u});
        add_equivalents(c, lo, hi);
        return true;
    }
    bool hexval(uint32_t ch, uint32_t* v) {
        if (ch >= '0' && ch <= '9') *v = ch - '0';
        else if (ch >= 'a' && ch <= 'f') *v = ch - 'a' + 10;
        else if (ch >= 'A' && ch <= 'F') *v = ch - 'A' + 10;
        else return false;
        return true;
    }
    bool hex_fixed(int digits, uint32_t* cp) {
        uint32_t v = 0, d;
        for (int k = 0; k < digits; ++k) {
            if (!more() || !hexval(peek(), &d)) return fail("bad hexadecimal escape");
            v = v * 16 + d;
            ++i;
        }
        *cp = v;
        return true;
    }
    bool hex_braced(uint32_t* cp) {  // {H..}
        ++i;
        uint32_t v = 0, d;
        int k = 0;
        while (more() && peek() != '}') {
            if (!hexval(peek(), &d) || ++k > 6) return fail("bad hexadecimal escape");
            v = v * 16 + d;
            ++i;
        }
        if (!more() || !k) return fail("bad hexadecimal escape");
        ++i;
        *cp = v;
        return true;
    }
    // \p{..} / \P{..}: General_Category values
    bool property(CharSet& c, bool ci, bool* negated) {
        const bool neg = peek() == 'P';
        ++i;
        std::string name;
        if (peek() == '{') {
            ++i;
            while (more() && peek() != '}') name += (char)(peek() < 128 ? peek() : '?'), ++i;
            if (!more()) return fail("unterminated \\p{");
            ++i;
        } else if (more()) {
            name += (char)peek();
            ++i;
        }
        bool inner_neg = false;
        if (!name.empty() && name[0] == '^') {
            inner_neg = true;
            name.erase(0, 1);
        }
        for (const char* pre : {"gc=", "General_Category=", "general_category="})
            if (name.rfind(pre, 0) == 0) name.erase(0, strlen(pre));
        static const struct { const char* a; const char* b; } alias[] = {
            {"Letter", "L"}, {"Mark", "M"}, {"Number", "N"}, {"Punctuation", "P"}, {"Symbol", "S"}, {"Separator", "Z"}, {"Other", "C"},
            {"Uppercase_Letter", "Lu"}, {"Lowercase_Letter", "Ll"}, {"Titlecase_Letter", "Lt"}, {"Modifier_Letter", "Lm"}, {"Other_Letter", "Lo"},
            {"Decimal_Number", "Nd"}, {"Letter_Number", "Nl"}, {"Other_Number", "No"}, {"Nonspacing_Mark", "Mn"}, {"Spacing_Mark", "Mc"},
            {"Enclosing_Mark", "Me"}, {"Control", "Cc"}, {"Format", "Cf"}, {"Unassigned", "Cn"}, {"Private_Use", "Co"}, {"Space_Separator", "Zs"}};
        for (const auto& a : alias)
            if (name == a.a) name = a.b;
        // (names are matched loosely, UTS #18: case, '_', '-' and blanks do not count -- \p{white_space}, \p{WHITE-SPACE})
        auto loose = [](const std::string& v) {
            std::string k;
            for (char ch : v)
                if (ch != '_' && ch != ' ' && ch != '-') k += (char)((ch >= 'A' && ch <= 'Z') ? ch + 32 : ch);
            return k;
        };
        const std::string lname = loose(name);
        if (name.size() > 2)  // (one- and two-letter names are General_Category values as they are spelled: \p{L}, \p{Lu}; "LC" below)
            for (const auto& a : alias)
                if (lname == loose(a.a)) name = a.b;
        if (lname == "whitespace" || lname == "space" || lname == "wspace") {  // == \s
            c.flags |= 0x20u;
            *negated = neg != inner_neg;
            return true;
        }
        uint32_t m = 0;
        if (name == "LC") m = 7u;
        else
            for (int g = 0; g < 30; ++g)
                if (name == GC_NAMES[g] || (name.size() == 1 && GC_NAMES[g][0] == name[0])) m |= 1u << g;
        if (!m) {  // a script?  Its ranges join the class (tk_regex_scripts.inc)
            std::string key;
            for (const char* pre : {"Script=", "script=", "sc=", "Is"})
                if (name.rfind(pre, 0) == 0 && name.size() > strlen(pre)) {
                    name.erase(0, strlen(pre));
                    break;
                }
            for (char ch : name)
                if (ch != '_' && ch != ' ' && ch != '-') key += (char)((ch >= 'A' && ch <= 'Z') ? ch + 32 : ch);
            for (const TkRxScript& sc : tk_rx_scripts) {
                const char* q = sc.names;
                while (*q) {
                    const char* e = strchr(q, '|');
                    const size_t len = e ? (size_t)(e - q) : strlen(q);
                    if (len == key.size() && !memcmp(q, key.data(), len)) {
                        for (unsigned k = 0; k < sc.cnt; ++k)
                            c.ranges.push_back({tk_rx_script_ranges[2 * (sc.off + k)], tk_rx_script_ranges[2 * (sc.off + k) + 1]});
                        *negated = neg != inner_neg;
                        return true;
                    }
                    q += len + (e ? 1 : 0);
                }
            }
            // a binary property (tk_regex_binprops.inc): categories that lie in it entirely + the ranges they leave out
            if (key == "any") {
                c.ranges.push_back({0u, 0x10FFFFu});
                *negated = neg != inner_neg;
                return true;
            }
            if (key == "ascii") {
                c.ranges.push_back({0u, 127u});
                *negated = neg != inner_neg;
                return true;
            }
            if (key == "assigned") {  // everything but Cn
                c.gcmask |= 0x3FFFFFFFu & ~(1u << 29);
                *negated = neg != inner_neg;
                return true;
            }
            for (const TkRxBinProp& bp : tk_rx_binprops) {
                const char* q = bp.names;
                while (*q) {
                    const char* e = strchr(q, '|');
                    const size_t len = e ? (size_t)(e - q) : strlen(q);
                    if (len == key.size() && !memcmp(q, key.data(), len)) {
                        if (ci && bp.case_sensitive) return fail("\\p{" + name + "} under (?i) is not supported");
                        c.gcmask |= bp.gcmask;
                        for (uint32_t cp = 0; cp < 128u; ++cp)
                            if ((bp.ascii[cp >> 5] >> (cp & 31u)) & 1u) c.ranges.push_back({cp, cp});
                        for (unsigned k = 0; k < bp.cnt; ++k)
                            c.ranges.push_back({tk_rx_binprop_ranges[2 * (bp.off + k)], tk_rx_binprop_ranges[2 * (bp.off + k) + 1]});
                        *negated = neg != inner_neg;
                        return true;
                    }
                    q += len + (e ? 1 : 0);
                }
            }
            return fail("\\p{" + name + "}: not a General_Category value, a script or a binary property of the UCD");
        }
        if (ci && (m & 7u) && (m & 7u) != 7u) return fail("\\p{" + name + "} under (?i) is not supported");
        c.gcmask |= m;
        *negated = neg != inner_neg;
        return true;
    }
    // an escape that stands for a set: fills `c`, *negated = the set is the complement
    bool class_escape(uint32_t e, CharSet& c, bool ci, bool* negated) {
        *negated = false;
        switch (e) {
            case 'd': c.gcmask |= 1u << 8; ++i; return true;
            case 'D': c.gcmask |= 1u << 8; *negated = true; ++i; return true;
            case 's': c.flags |= 0x20u; ++i; return true;
            case 'S': c.flags |= 0x20u; *negated = true; ++i; return true;
            case 'w': c.flags |= 0x40u; ++i; return true;
            case 'W': c.flags |= 0x40u; *negated = true; ++i; return true;
            // fancy-regex: \h = hex digit [0-9A-Fa-f], \H = [^0-9A-Fa-f] (Oniguruma's meaning -- not PCRE's horizontal white space).  Both as plain
            // ranges (the complement of three ranges is four), so that they may stand inside a class as well; closed under simple case folding
            // as they are, (?i) changes nothing
            case 'h': c.ranges.insert(c.ranges.end(), {{'0', '9'}, {'A', 'F'}, {'a', 'f'}}); ++i; return true;
            case 'H': c.ranges.insert(c.ranges.end(), {{0u, '0' - 1u}, {'9' + 1u, 'A' - 1u}, {'F' + 1u, 'a' - 1u}, {'f' + 1u, 0x10FFFFu}}); ++i; return true;
            default: return property(c, ci, negated);
        }
    }
    static bool is_class_escape(uint32_t e) {
        return e == 'd' || e == 'D' || e == 's' || e == 'S' || e == 'w' || e == 'W' || e == 'p' || e == 'P' || e == 'h' || e == 'H';
    }
    bool scalar(uint32_t cp) {
        if (cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) return fail("escape is not a Unicode scalar value");
        return true;
    }
    // a literal escape behind the backslash (i at the escape letter): one code point
    bool literal_escape(uint32_t* cp) {
        const uint32_t e = peek();
        ++i;
        switch (e) {
            case 'n': *cp = '\n'; return true;
            case 'r': *cp = '\r'; return true;
            case 't': *cp = '\t'; return true;
            case 'f': *cp = '\f'; return true;
            case 'v': *cp = '\v'; return true;
            case 'a': *cp = 7; return true;
            case 'e': *cp = 27; return true;
            case '0': *cp = 0; return true;
            case 'x': return (peek() == '{' ? hex_braced(cp) : hex_fixed(2, cp)) && scalar(*cp);
            case 'u': return (peek() == '{' ? hex_braced(cp) : hex_fixed(4, cp)) && scalar(*cp);
            case 'U': return hex_fixed(8, cp) && scalar(*cp);
            default: break;
        }
        if (e >= '1' && e <= '9') return fail("back-references are not supported");
        if (e == 'b' || e == 'B') return fail("\\b inside a class is not supported");
        if (e == 'G' || e == 'K' || e == 'Z' || e == 'k' || e == 'g' || e == 'X' || e == 'R' || e == 'N')
            return fail(std::string("the escape \\") + (char)e + " is not supported");
        if ((e >= 'a' && e <= 'z') || (e >= 'A' && e <= 'Z')) return fail(std::string("unknown escape \\") + (char)e);
        *cp = e;  // escaped punctuation
        return true;
    }
    // [:name:] / [:^name:] inside a class (i at '['): the ASCII classes of the Rust `regex` crate
    bool posix_class(CharSet& c, bool ci) {
        size_t j = i + 2;
        bool neg = false;
        if (j < s.size() && s[j] == '^') {
            neg = true;
            ++j;
        }
        std::string name;
        while (j < s.size() && s[j] >= 'a' && s[j] <= 'z') name += (char)s[j++];
        if (j + 1 >= s.size() || s[j] != ':' || s[j + 1] != ']') return fail("malformed POSIX class (expected [:name:])");
        static const struct { const char* name; const char* set; } posix[] = {  // pairs lo, hi
            {"alnum", "09AZaz"}, {"alpha", "AZaz"}, {"ascii", "\x01\x7f"}, {"blank", "\t\t  "}, {"cntrl", "\x01\x1f\x7f\x7f"}, {"digit", "09"},
            {"graph", "!~"}, {"lower", "az"}, {"print", " ~"}, {"punct", "!/:@[`{~"}, {"space", "\t\r  "}, {"upper", "AZ"},
            {"word", "09AZaz__"}, {"xdigit", "09AFaf"}};
        for (const auto& pc : posix) {
            if (name != pc.name) continue;
            bool in[128] = {false};
            for (const char* q = pc.set; *q; q += 2)
                for (int cp = q[0]; cp <= q[1]; ++cp) in[cp] = true;
            if (name == "ascii" || name == "cntrl") in[0] = true;  // (NUL cannot stand in the string above)
            if (!neg) {
                for (uint32_t cp = 0; cp < 128u; ++cp)
                    if (in[cp] && !add_char(c, cp, ci)) return false;
            } else {
                for (uint32_t cp = 0; cp < 128u; ++cp)
                    if (!in[cp] && !add_char(c, cp, ci)) return false;
                c.ranges.push_back({128u, 0x10FFFFu});
            }
            i = j + 2;
            return true;
        }
        return fail("unknown POSIX class [:" + name + ":]");
    }
    int parse_class(const Flags& f) {  // i at '['
        CharSet c;
        if (!parse_class_body(f, &c, true)) return -1;
        return add_set(c);
    }
    // [ ... ] -> *out.  Set operations (top level of a class only): [A&&B] intersection, [A--B] difference, B a class or an escape; chains.
    bool parse_class_body(const Flags& f, CharSet* out, bool allow_ops) {
        ++i;
        CharSet c;
        if (peek() == '^') {
            c.neg = true;
            ++i;
        }
        bool first = true;
        int tail = -1;  // last operand of a chain of set operations (index into sets), -1: still the left side
        while (more() && (peek() != ']' || first)) {
            first = false;
            uint32_t lo;
            const bool op_and = peek() == '&' && peek(1) == '&', op_diff = peek() == '-' && peek(1) == '-';
            if (op_and || op_diff) {
                if (!allow_ops) return fail("a set operation inside the operand of a set operation is not supported");
                i += 2;
                CharSet operand;
                if (peek() == '[') {
                    if (!parse_class_body(f, &operand, false)) return false;
                } else if (peek() == '\\' && is_class_escape(peek(1))) {
                    ++i;
                    bool negated;
                    if (!class_escape(peek(), operand, f.ci, &negated)) return false;
                    operand.neg = negated;
                } else {
                    return fail("the right side of a class set operation has to be a class or an escape like \\p{..}");
                }
                if (op_diff) operand.neg = !operand.neg;
                sets.push_back(operand);
                const int idx = (int)sets.size() - 1;
                if (tail < 0) c.and_set = idx;
                else sets[tail].and_set = idx;
                tail = idx;
                if (peek() != ']' && !(peek() == '&' && peek(1) == '&') && !(peek() == '-' && peek(1) == '-'))
                    return fail("only ']' or another set operation may follow the operand of a set operation");
                continue;
            }
            if (peek() == '[') {
                if (peek(1) == ':') {  // [:alpha:] / [:^alpha:] -- ASCII, as in the Rust `regex` crate
                    if (!posix_class(c, f.ci)) return false;
                    continue;
                }
                return fail("nested classes are only supported as operands of && and --");
            }
            if (peek() == '~' && peek(1) == '~') return fail("the class set operation ~~ is not supported");
            if (peek() == '\\') {
                ++i;
                if (!more()) return fail("pattern ends in a backslash");
                if (is_class_escape(peek())) {
                    bool negated;
                    CharSet sub;
                    if (!class_escape(peek(), sub, f.ci, &negated)) return false;
                    if (negated) {  // a complement inside a union: one per class ([^\S\n], [\S\d])
                        if (c.comp) return fail("more than one negated escape inside a class is not supported");
                        if (!sub.ranges.empty()) return fail("a negated script inside a class is not supported (negate the class: [^\\p{..}])");
                        c.comp = true;
                        c.cgcmask = sub.gcmask;
                        c.cflags = sub.flags;
                    } else {
                        c.gcmask |= sub.gcmask;
                        c.flags |= sub.flags;
                        c.ranges.insert(c.ranges.end(), sub.ranges.begin(), sub.ranges.end());
                    }
                    continue;
                }
                if (!literal_escape(&lo)) return false;
            } else {
                lo = peek();
                ++i;
            }
            if (peek() == '-' && peek(1) != ']' && peek(1) != '-' && i + 1 < s.size()) {  // range
                ++i;
                uint32_t hi;
                if (peek() == '\\') {
                    ++i;
                    if (!more() || is_class_escape(peek())) return fail("bad class range");
                    if (!literal_escape(&hi)) return false;
                } else {
                    hi = peek();
                    ++i;
                }
                if (!add_range(c, lo, hi, f.ci)) return false;
            } else if (!add_char(c, lo, f.ci)) {
                return false;
            }
        }
        if (!more()) return fail("unterminated class");
        ++i;  // ]
        *out = c;
        return true;
    }

    // the node as a fixed-length sequence of sets (set indices appended to *seq): sets, concatenations, x{m} of such; false otherwise
    bool behind_sequence(int n, std::vector<int>* seq) const {
        const Node& N = nodes[n];
        if (N.kind == Node::SET) {
            seq->push_back(N.set);
            return true;
        }
        if (N.kind == Node::CAT) {
            for (int k : N.kids)
                if (!behind_sequence(k, seq)) return false;
            return true;
        }
        if (N.kind == Node::REPEAT && N.mn == N.mx && N.mn <= 16) {
            for (uint32_t k = 0; k < N.mn; ++k)
                if (!behind_sequence(N.kids[0], seq) || seq->size() > 16) return false;
            return true;
        }
        return false;
    }

    // ---- expressions
    int parse_alt(Flags f, int depth) {
        if (depth > 40) return fail("pattern nested too deeply"), -1;
        std::vector<int> alts;
        for (;;) {
            const int c = parse_cat(f, depth);
            if (c < 0) return -1;
            alts.push_back(c);
            if (peek() == '|' && more()) {
                ++i;
                continue;
            }
            break;
        }
        if (alts.size() == 1) return alts[0];
        Node n;
        n.kind = Node::ALT;
        n.kids = alts;
        return add(n);
    }
    int parse_cat(Flags& f, int depth) {  // (inline flags (?i) change f for the rest of the enclosing group)
        Node cat;
        cat.kind = Node::CAT;
        for (;;) {
            skip_verbose(f);
            if (!more() || peek() == '|' || peek() == ')') break;
            int a = parse_atom(f, depth);
            if (a == -2) continue;  // inline flags
            if (a < 0) return -1;
            skip_verbose(f);
            a = parse_quant(a);
            if (a < 0) return -1;
            cat.kids.push_back(a);
        }
        if (cat.kids.size() == 1) return cat.kids[0];
        if (cat.kids.empty()) cat.kind = Node::EMPTY;
        return add(cat);
    }
    // (?x): white space and #-comments between the tokens of the pattern mean nothing (inside a class they stay literal)
    void skip_verbose(const Flags& f) {
        while (f.verbose && more()) {
            const uint32_t c = peek();
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') {
                ++i;
            } else if (c == '#') {
                while (more() && peek() != '\n') ++i;
            } else {
                break;
            }
        }
    }
    int parse_quant(int a) {
        for (;;) {
            uint32_t mn, mx;
            const uint32_t q = peek();
            if (!more()) return a;
            if (q == '?') mn = 0, mx = 1, ++i;
            else if (q == '*') mn = 0, mx = TK_RX_INF, ++i;
            else if (q == '+') mn = 1, mx = TK_RX_INF, ++i;
            else if (q == '{') {
                size_t j = i + 1;
                uint64_t v = 0;
                int nd = 0;
                while (j < s.size() && s[j] >= '0' && s[j] <= '9' && nd < 6) v = v * 10 + (s[j] - '0'), ++j, ++nd;
                if (!nd) return fail("bad quantifier"), -1;
                mn = mx = (uint32_t)v;
                if (j < s.size() && s[j] == ',') {
                    ++j;
                    v = 0, nd = 0;
                    while (j < s.size() && s[j] >= '0' && s[j] <= '9' && nd < 6) v = v * 10 + (s[j] - '0'), ++j, ++nd;
                    mx = nd ? (uint32_t)v : TK_RX_INF;
                }
                if (j >= s.size() || s[j] != '}') return fail("bad quantifier"), -1;
                if (mx < mn) return fail("quantifier range out of order"), -1;
                i = j + 1;
            } else {
                return a;
            }
            const Node::Kind k = nodes[a].kind;
            if (k == Node::START || k == Node::END || k == Node::LOOK || k == Node::EMPTY || k == Node::WORDB || k == Node::BEHIND) return fail("nothing to repeat"), -1;
            Node r;
            r.kind = Node::REPEAT;
            r.kids = {a};
            r.mn = mn;
            r.mx = mx;
            if (peek() == '?' && more()) r.mode = TK_RX_LAZY, ++i;
            else if (peek() == '+' && more()) r.mode = TK_RX_POSSESSIVE, ++i;
            a = add(r);
            if (more() && (peek() == '*' || peek() == '+' || peek() == '?' || peek() == '{')) return fail("a quantifier behind a quantifier"), -1;
        }
    }
    int parse_atom(Flags& f, int depth) {
        const uint32_t c = peek();
        if (c == '(') {
            ++i;
            Flags g = f;
            Node::Kind wrap = Node::EMPTY;
            bool neg = false;
            if (peek() == '?') {
                ++i;
                const uint32_t k = peek();
                if (k == ':') ++i;
                else if (k == '>') wrap = Node::ATOMIC, ++i;
                else if (k == '=') wrap = Node::LOOK, ++i;
                else if (k == '!') wrap = Node::LOOK, neg = true, ++i;
                else if (k == '<' && (peek(1) == '=' || peek(1) == '!')) {  // look-behind: fixed-length sequences of single chars
                    const bool negative = peek(1) == '!';
                    i += 2;
                    const int body = parse_alt(g, depth + 1);
                    if (body < 0) return -1;
                    if (peek() != ')' || !more()) return fail("unterminated group"), -1;
                    ++i;
                    std::vector<int> seq;
                    const Node& B = nodes[body];
                    const std::vector<int> alts = B.kind == Node::ALT ? B.kids : std::vector<int>{body};
                    for (int a : alts) {
                        seq.clear();
                        if (!behind_sequence(a, &seq) || seq.empty() || seq.size() > 16)
                            return fail("look-behind has to be an alternation of fixed-length sequences (at most 16 chars) of single chars or classes"), -1;
                    }
                    Node n;
                    n.kind = Node::BEHIND;
                    n.kids = {body};
                    n.neg = negative;
                    return add(n);
                }
                else if (k == 'P' || k == '<' || k == '\'') {  // named group: a plain group
                    const uint32_t close = k == '\'' ? '\'' : '>';
                    if (k == 'P') ++i;
                    if (peek() == '=' || peek() == '>') return fail("named back-references are not supported"), -1;
                    ++i;
                    while (more() && peek() != close) ++i;
                    if (!more()) return fail("unterminated group name"), -1;
                    ++i;
                } else {  // flags: (?i) (?s) (?is:...) (?-i)
                    bool on = true, any = false;
                    for (;; ++i) {
                        const uint32_t fl = peek();
                        if (fl == '-') on = false;
                        else if (fl == 'i') g.ci = on, any = true;
                        else if (fl == 's') g.dotall = on, any = true;
                        else if (fl == 'm') g.multiline = on, any = true;
                        else if (fl == 'x') g.verbose = on, any = true;
                        else if (fl == 'u') any = true;
                        else if (fl == 'U' || fl == 'R') return fail(std::string("the flag (?") + (char)fl + ") is not supported"), -1;
                        else break;
                    }
                    if (!any) return fail("unknown group syntax"), -1;
                    if (peek() == ')') {  // inline: for the rest of the enclosing group
                        ++i;
                        f = g;
                        return -2;
                    }
                    if (peek() != ':') return fail("unknown group syntax"), -1;
                    ++i;
                }
            }
            int body = parse_alt(g, depth + 1);
            if (body < 0) return -1;
            if (peek() != ')' || !more()) return fail("unterminated group"), -1;
            ++i;
            if (wrap == Node::EMPTY) return body;
            Node n;
            n.kind = wrap;
            n.kids = {body};
            n.neg = neg;
            return add(n);
        }
        if (c == '[') return parse_class(f);
        if (c == '.') {
            ++i;
            CharSet cs;
            cs.neg = true;
            if (!f.dotall) cs.ranges.push_back({'\n', '\n'});
            return add_set(cs);
        }
        if (c == '^' || c == '$') {
            ++i;
            Node n;
            n.kind = c == '^' ? Node::START : Node::END;
            const int anchor = add(n);
            if (!f.multiline) return anchor;
            // (?m): ^ also behind a newline, $ also in front of one -- (?:\A|(?<=\n)), (?:\z|(?=\n))
            CharSet nl;
            nl.ranges.push_back({'\n', '\n'});
            const int set_node = add_set(nl);
            Node look;
            look.kind = c == '^' ? Node::BEHIND : Node::LOOK;
            look.kids = {set_node};
            const int look_node = add(look);
            Node alt;
            alt.kind = Node::ALT;
            alt.kids = {anchor, look_node};
            return add(alt);
        }
        if (c == '*' || c == '+' || c == '?') return fail("nothing to repeat"), -1;
        if (c == '{' || c == '}' || c == ']') return fail(std::string("unescaped '") + (char)c + "'"), -1;
        if (c == '\\') {
            ++i;
            if (!more()) return fail("pattern ends in a backslash"), -1;
            const uint32_t e = peek();
            if (e == 'A' || e == 'z') {
                ++i;
                Node n;
                n.kind = e == 'A' ? Node::START : Node::END;
                return add(n);
            }
            if (e == 'O') {  // fancy-regex: any char, newline included, whatever (?s) says
                ++i;
                CharSet cs;
                cs.neg = true;
                return add_set(cs);
            }
            if (e == 'b' || e == 'B') {
                ++i;
                Node n;
                n.kind = Node::WORDB;
                n.neg = e == 'B';
                return add(n);
            }
            if (is_class_escape(e)) {
                CharSet cs;
                bool negated;
                if (!class_escape(e, cs, f.ci, &negated)) return -1;
                cs.neg = negated;
                return add_set(cs);
            }
            uint32_t cp;
            if (!literal_escape(&cp)) return -1;
            CharSet cs;
            if (!add_char(cs, cp, f.ci)) return -1;
            return add_set(cs);
        }
        ++i;
        CharSet cs;
        if (!add_char(cs, c, f.ci)) return -1;
        return add_set(cs);
    }
};

// emit-time mark on a SPLIT whose frame a later POP removes: first-byte pruning may skip its first choice but must not drop the frame
const uint32_t SPLIT_KEEPS_FRAME = 0x80000000u;

struct Emitter {
    const Parser& P;
    std::vector<TkRxIns> code;
    std::string err;
    explicit Emitter(const Parser& p) : P(p) {}

    uint32_t minlen(int n) const {
        const Node& N = P.nodes[n];
        switch (N.kind) {
            case Node::SET: return 1;
            case Node::CAT: {
                uint64_t t = 0;
                for (int k : N.kids) t += minlen(k);
                return t > 0xFFFFFF ? 0xFFFFFFu : (uint32_t)t;
            }
            case Node::ALT: {
                uint32_t m = 0xFFFFFFFFu;
                for (int k : N.kids) m = m < minlen(k) ? m : minlen(k);
                return m;
            }
            case Node::REPEAT: {
                const uint64_t t = (uint64_t)minlen(N.kids[0]) * N.mn;
                return t > 0xFFFFFF ? 0xFFFFFFu : (uint32_t)t;
            }
            case Node::ATOMIC: return minlen(N.kids[0]);
            default: return 0;
        }
    }
    uint32_t here() const { return (uint32_t)code.size(); }
    uint32_t put(uint32_t op, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0) {
        code.push_back(TkRxIns{op, a, b, c});
        return here() - 1;
    }
    // tail: nothing that can fail follows this node inside its alternative (a repeated group then needs no way back into it)
    bool emit(int n, bool tail) {
        if (code.size() > 4096) {
            if (err.empty()) err = "the pattern is too large";
            return false;
        }
        const Node& N = P.nodes[n];
        switch (N.kind) {
            case Node::EMPTY: return true;
            case Node::SET: put(TK_RX_SET, (uint32_t)N.set); return true;
            case Node::START: put(TK_RX_START); return true;
            case Node::WORDB: put(TK_RX_WORDB, N.neg ? 1u : 0u); return true;
            case Node::END: put(TK_RX_END); return true;
            case Node::CAT:
                for (size_t k = 0; k < N.kids.size(); ++k)
                    if (!emit(N.kids[k], tail && k + 1 == N.kids.size())) return false;
                return true;
            case Node::ALT: {
                std::vector<uint32_t> jumps;
                for (size_t k = 0; k < N.kids.size(); ++k) {
                    uint32_t split = 0;
                    const bool last = k + 1 == N.kids.size();
                    if (!last) split = put(TK_RX_SPLIT);
                    if (!last) code[split].a = here();
                    if (!emit(N.kids[k], tail)) return false;
                    if (!last) {
                        jumps.push_back(put(TK_RX_JMP));
                        code[split].b = here();
                    }
                }
                for (uint32_t j : jumps) code[j].a = here();
                return true;
            }
            case Node::ATOMIC: {
                put(TK_RX_ATOM_BEGIN);
                if (!emit(N.kids[0], true)) return false;
                put(TK_RX_ATOM_END);
                return true;
            }
            case Node::LOOK: {
                const uint32_t b = put(TK_RX_LOOK_BEGIN, N.neg ? 1u : 0u);
                if (!emit(N.kids[0], true)) return false;
                put(TK_RX_LOOK_END);
                code[b].b = here();
                return true;
            }
            case Node::BEHIND: {  // a zero-width group like a look-ahead; each alternative: its chars, the nearest last, as PREV tests
                const uint32_t b = put(TK_RX_LOOK_BEGIN, N.neg ? 1u : 0u);
                const Node& B = P.nodes[N.kids[0]];
                const std::vector<int> alts = B.kind == Node::ALT ? B.kids : std::vector<int>{N.kids[0]};
                std::vector<uint32_t> jumps;
                for (size_t k = 0; k < alts.size(); ++k) {
                    const bool last = k + 1 == alts.size();
                    uint32_t split = 0;
                    if (!last) {
                        split = put(TK_RX_SPLIT);
                        code[split].a = here();
                    }
                    std::vector<int> seq;
                    P.behind_sequence(alts[k], &seq);
                    for (size_t j = 0; j < seq.size(); ++j) put(TK_RX_PREV, (uint32_t)seq[j], 0u, (uint32_t)(seq.size() - j));
                    if (!last) {
                        jumps.push_back(put(TK_RX_JMP));
                        code[split].b = here();
                    }
                }
                for (uint32_t j : jumps) code[j].a = here();
                put(TK_RX_LOOK_END);
                code[b].b = here();
                return true;
            }
            case Node::REPEAT: return emit_repeat(N, tail);
        }
        return true;
    }
    bool emit_repeat(const Node& N, bool tail) {
        const int body = N.kids[0];
        const Node& B = P.nodes[body];
        if (N.mx == 0) return true;  // x{0}
        if (B.kind == Node::SET) {    // one instruction, one backtrack frame
            put(TK_RX_REP | ((uint32_t)N.mode << 8), (uint32_t)B.set, N.mn, N.mx);
            return true;
        }
        // (bounded repeats as well: what an iteration that matches nothing means for the ones behind it differs between engines -- Python `regex`
        // leaves the loop, fancy-regex's VM fails the path, a plain backtracker carries on -- and the crate's source is not here to pin it:
        // (?:\pL?+(?!\d)|(\s\p{Lu}){2}){1,3}\w+? on "\nS\nKcd" is one piece or two.  Found by tools/fuzz_regex.py, offset 55.  x? is fine.)
        if (N.mx >= 2 && minlen(body) == 0) {
            err = "a repeated group that can match the empty string is not supported";
            return false;
        }
        if (N.mn > 64 || (N.mx != TK_RX_INF && N.mx > 64)) {
            err = "a group repeated more than 64 times is not supported";
            return false;
        }
        const bool lazy = N.mode == TK_RX_LAZY;
        if (N.mx == TK_RX_INF && !lazy && (N.mode == TK_RX_POSSESSIVE || tail)) {
            // No way back into a finished repetition is ever taken once the minimum is reached (possessive; or greedy where nothing can
            // fail behind the loop: the way out of the last repetition leads to a match): every further repetition is atomic and forgets
            // the previous one's way out -- a constant number of frames however often the group repeats.  The first `mn` repetitions keep
            // their alternatives (x{2,}: if the second cannot match, the first may have to give something back).
            const bool poss = N.mode == TK_RX_POSSESSIVE;
            if (poss) put(TK_RX_ATOM_BEGIN);
            for (uint32_t k = 0; k < N.mn; ++k)
                if (!emit(body, false)) return false;
            const uint32_t sp = put(TK_RX_SPLIT | SPLIT_KEEPS_FRAME);  // (the POP below takes this split's frame: it has to be there)
            code[sp].a = here();
            put(TK_RX_ATOM_BEGIN);
            if (!emit(body, true)) return false;
            put(TK_RX_ATOM_END);
            put(TK_RX_POP);
            put(TK_RX_JMP, sp);
            code[sp].b = here();
            if (poss) put(TK_RX_ATOM_END);
            return true;
        }
        const bool poss = N.mode == TK_RX_POSSESSIVE;
        if (poss) put(TK_RX_ATOM_BEGIN);
        for (uint32_t k = 0; k < N.mn; ++k)
            if (!emit(body, false)) return false;
        // every split prefers the body (greedy) or the way out (lazy)
        if (N.mx == TK_RX_INF) {  // L: split(body, out); body; jmp L
            const uint32_t sp = put(TK_RX_SPLIT);
            const uint32_t body_at = here();
            if (!emit(body, false)) return false;
            put(TK_RX_JMP, sp);
            const uint32_t out = here();
            code[sp].a = lazy ? out : body_at;
            code[sp].b = lazy ? body_at : out;
        } else {  // (body (body (body)?)?)?: every split leaves for the common end
            std::vector<uint32_t> splits;
            for (uint32_t k = N.mn; k < N.mx; ++k) {
                const uint32_t sp = put(TK_RX_SPLIT);
                splits.push_back(sp);
                code[sp].a = here();  // (patched below for lazy)
                if (!emit(body, false)) return false;
            }
            const uint32_t out = here();
            for (uint32_t sp : splits) {
                const uint32_t body_at = code[sp].a;
                code[sp].a = lazy ? out : body_at;
                code[sp].b = lazy ? body_at : out;
            }
        }
        if (poss) put(TK_RX_ATOM_END);
        return true;
    }
};

bool utf8_to_cps(const char* p, std::vector<uint32_t>* out) {
    const uint8_t* s = (const uint8_t*)p;
    while (*s) {
        uint32_t b0 = *s, need = b0 < 0x80 ? 1 : (b0 >= 0xF0 ? 4 : (b0 >= 0xE0 ? 3 : (b0 >= 0xC0 ? 2 : 0)));
        if (!need) return false;
        uint32_t cp = need == 1 ? b0 : (b0 & (0x7Fu >> need));
        for (uint32_t k = 1; k < need; ++k) {
            if ((s[k] & 0xC0u) != 0x80u) return false;
            cp = (cp << 6) | (s[k] & 0x3Fu);
        }
        out->push_back(cp);
        s += need;
    }
    return true;
}

// ---- first-byte bitmaps.  first(i) = the bytes a match of the program from instruction i can begin with (before any byte is consumed),
// as the least fixpoint of the obvious equations; whatever cannot be known is "every byte" (reaching MATCH or the end of a look-ahead
// succeeds whatever follows, and so -- for this purpose -- does reaching the end of an atomic group; a look-ahead in front is ignored:
// it only restricts).  A SPLIT then carries the bitmaps of its two targets:
// the matcher does not take -- and does not keep as a way back -- a choice that cannot begin with the byte at the position, which it would
// find out by failing a few instructions later.  (Not consulted at the end of the haystack, where `$` and empty tails decide.)
struct Bits256 {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool merge(const Bits256& o) {
        bool ch = false;
        for (int k = 0; k < 8; ++k) {
            const uint32_t v = w[k] | o.w[k];
            ch |= v != w[k];
            w[k] = v;
        }
        return ch;
    }
    void set(uint32_t b) { w[b >> 5] |= 1u << (b & 31u); }
    void all() {
        for (uint32_t& x : w) x = 0xFFFFFFFFu;
    }
    bool operator==(const Bits256& o) const { return !memcmp(w, o.w, sizeof w); }
};

uint32_t utf8_lead(uint32_t cp) { return cp < 0x80u ? cp : (cp < 0x800u ? 0xC0u | (cp >> 6) : (cp < 0x10000u ? 0xE0u | (cp >> 12) : 0xF0u | (cp >> 18))); }

Bits256 first_bytes_of_set(const TkRxSet& S, const std::vector<uint32_t>& ranges) {
    Bits256 b;
    for (uint32_t c = 0; c < 128; ++c)
        if ((S.ascii[c >> 5] >> (c & 31u)) & 1u) b.set(c);
    const uint32_t roff = S.rr >> 16, rcnt = S.rr & 0xFFFFu;
    if ((S.flags & 3u) || S.gcmask || (S.flags & 0x60u)) {  // negated / complemented / property members: any byte from 0x80 on
        for (uint32_t c = 128; c < 256; ++c) b.set(c);       // (malformed bytes read as U+FFFD, which such a set may hold)
        return b;
    }
    for (uint32_t i = 0; i < rcnt; ++i) {
        const uint32_t lo = ranges[2 * (roff + i)], hi = ranges[2 * (roff + i) + 1];
        for (uint32_t c = utf8_lead(lo); c <= utf8_lead(hi > 0x10FFFFu ? 0x10FFFFu : hi); ++c) b.set(c);
        if (lo <= 0xFFFDu && hi >= 0xFFFDu)
            for (uint32_t c = 128; c < 256; ++c) b.set(c);
    }
    return b;
}

void annotate_first_bytes(TkRxCompiled* out) {
    std::vector<TkRxIns>& code = out->ins;
    if (getenv("TIKTOKEN_AMD_RX_NO_PRUNING")) {  // (tests: the same program without the bitmaps must split the same way)
        for (TkRxIns& I : code)
            if ((I.op & 0xFFu) == TK_RX_SPLIT) I.op = TK_RX_SPLIT;
        out->first.assign(8, 0xFFFFFFFFu);
        return;
    }
    const size_t n = code.size();
    std::vector<Bits256> first(n), of_set(out->sets.size());
    for (size_t s = 0; s < out->sets.size(); ++s) of_set[s] = first_bytes_of_set(out->sets[s], out->ranges);
    Bits256 every;
    every.all();
    for (bool changed = true; changed;) {
        changed = false;
        for (size_t i = n; i-- > 0;) {
            const TkRxIns& I = code[i];
            Bits256 f;
            auto at = [&](uint32_t j) -> const Bits256& { return j < n ? first[j] : every; };
            switch (I.op & 0xFFu) {
                case TK_RX_SET: f = of_set[I.a]; break;
                case TK_RX_REP:
                    f = of_set[I.a];
                    if (I.b == 0) f.merge(at((uint32_t)i + 1));
                    break;
                case TK_RX_SPLIT:
                    f = at(I.a);
                    f.merge(at(I.b));
                    break;
                case TK_RX_JMP: f = at(I.a); break;
                case TK_RX_MATCH:
                case TK_RX_LOOK_END:
                case TK_RX_ATOM_END:  // (a commit point: failing before it may still use the group's alternatives, failing behind it may not --
                case TK_RX_POP:       //  pruning must not look across)
                    f = every;
                    break;
                case TK_RX_END:
                case TK_RX_FAIL: break;  // (nothing: `$` fails unless the haystack ends here, and there the bitmaps are not consulted)
                case TK_RX_LOOK_BEGIN: f = at(I.b); break;
                default: f = at((uint32_t)i + 1); break;  // START, ATOM_BEGIN: on to the next instruction
            }
            changed |= first[i].merge(f);
        }
    }
    std::vector<Bits256> uniq;
    auto id_of = [&](uint32_t target) -> uint32_t {  // 1 + index, 0 = no bitmap (every byte, or the table is full)
        const Bits256& f = target < n ? first[target] : every;
        if (f == every) return 0;
        for (size_t k = 0; k < uniq.size(); ++k)
            if (uniq[k] == f) return (uint32_t)k + 1;
        if (uniq.size() >= TK_RX_MAX_FIRST) return 0;
        uniq.push_back(f);
        return (uint32_t)uniq.size();
    };
    for (TkRxIns& I : code)
        if ((I.op & 0xFFu) == TK_RX_SPLIT) I.op = TK_RX_SPLIT | id_of(I.a) << 8 | ((I.op & SPLIT_KEEPS_FRAME) ? 0u : id_of(I.b) << 16);
    out->first.clear();
    for (const Bits256& f : uniq) out->first.insert(out->first.end(), f.w, f.w + 8);
    if (out->first.empty()) out->first.assign(8, 0xFFFFFFFFu);  // (never indexed; keeps the upload simple)
}

#include "tk_regex_dfa.inc"

}  // namespace

const uint8_t* tk_rx_props_stage1() { return tk_rx_stage1; }
const uint8_t* tk_rx_props_stage2() { return tk_rx_stage2; }
uint32_t tk_rx_props_blocks() { return TK_RX_NBLOCKS; }

std::string tk_rx_compile(const char* pat_str, TkRxCompiled* out) {
    Parser P;
    if (!utf8_to_cps(pat_str, &P.s)) return "the pattern is not valid UTF-8";
    Flags f;
    const int root = P.parse_alt(f, 0);
    if (root < 0) return P.err.empty() ? "cannot parse the pattern" : P.err;
    if (P.more()) return P.peek() == ')' ? "unbalanced ')'" : "cannot parse the pattern";
    Emitter E(P);
    if (E.minlen(root) == 0) return "the pattern can match the empty string (a piece must hold at least one char)";
    if (!E.emit(root, true)) return E.err;
    E.put(TK_RX_MATCH);
    if (E.code.size() > TK_RX_MAX_INS) return "the pattern is too large (more than " + std::to_string(TK_RX_MAX_INS) + " instructions)";
    if (P.sets.size() > TK_RX_MAX_SETS) return "the pattern has too many classes";
    out->ins = E.code;
    out->sets.clear();
    out->ranges.clear();
    for (const CharSet& c : P.sets) {
        TkRxSet S{};
        for (uint32_t cp = 0; cp < 128; ++cp)
            if (P.member(c, cp)) S.ascii[cp >> 5] |= 1u << (cp & 31u);
        S.gcmask = c.gcmask;
        S.cgcmask = c.cgcmask;
        S.flags = (c.flags & 0x60u) | (c.neg ? 1u : 0u) | (c.comp ? 2u : 0u) | ((c.cflags & 0x60u) << 8) | ((uint32_t)(c.and_set + 1) << 16);
        const uint32_t roff = (uint32_t)out->ranges.size() / 2;
        std::vector<std::pair<uint32_t, uint32_t>> rs;  // beyond ASCII (the ASCII part lives in the bitmap), sorted and merged: the matcher
        for (const auto& r : c.ranges)                   // looks a code point up by bisection
            if (r.second >= 128u) rs.push_back({r.first < 128u ? 128u : r.first, r.second});
        std::sort(rs.begin(), rs.end());
        for (const auto& r : rs) {
            const size_t k = out->ranges.size();
            if (k > 2 * (size_t)roff && r.first <= out->ranges[k - 1] + 1u) {
                if (r.second > out->ranges[k - 1]) out->ranges[k - 1] = r.second;
            } else {
                out->ranges.push_back(r.first);
                out->ranges.push_back(r.second);
            }
        }
        S.rr = roff << 16 | ((uint32_t)out->ranges.size() / 2 - roff);
        // \p{X}+|\P{X}+ : two sets, one list of ranges (a binary property can be several hundred of them)
        for (const TkRxSet& Q : out->sets) {
            const uint32_t qo = Q.rr >> 16, qc = Q.rr & 0xFFFFu, cnt = S.rr & 0xFFFFu;
            if (cnt && qc == cnt && qo != roff && std::equal(out->ranges.begin() + 2 * qo, out->ranges.begin() + 2 * (qo + qc), out->ranges.begin() + 2 * roff)) {
                out->ranges.resize(2 * (size_t)roff);
                S.rr = qo << 16 | qc;
                break;
            }
        }
        out->sets.push_back(S);
    }
    if (out->ranges.size() / 2 > TK_RX_MAX_RANGES) return "the pattern has too many class ranges";
    annotate_first_bytes(out);
    build_dfa(P, root, out);  // the table form of the same pattern, where it has one (tk_regex_dfa.inc)
    return "";
}

TkRxProg TkRxCompiled::view() const {
    TkRxProg P{ins.data(), sets.data(), ranges.data(), tk_rx_stage1, tk_rx_stage2, (uint32_t)ins.size(), (uint32_t)sets.size(),
               (uint32_t)ranges.size() / 2, first.data(), (uint32_t)first.size() / 8};
    if (has_dfa()) {
        P.dfa_trans = dfa_trans.data();
        P.dfa_ascii = dfa_ascii.data();
        P.dfa_s1 = dfa_s1.data();
        P.dfa_s2 = dfa_s2.data();
        P.dfa_ncls = dfa_ncls;
        P.dfa_flags = dfa_flags;
    }
    return P;
}
