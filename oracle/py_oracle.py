"""CPU parity oracle (Python) for the BPE encode hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product path (tiktoken_amd/) may import this module; only tests/,
tools/ (fixture generation), __graft_entry__.smoke() and bench.py's cpu_baseline
leg use it, and only as the checker.

What is restated here, and from where (all citations into /root/reference):

* `split_regex`        -- the sanctioned "regex in Python + native BPE" shape,
                          tiktoken/core.py:395-404 and tiktoken/_educational.py:23-37:
                          `regex.findall(pat_str, text)`.
* `split_scan`         -- a sequential scanner for the three stock `pat_str`s
                          (tiktoken_ext/openai_public.py:12-14, :89, :104-114) that
                          yields what fancy-regex `find_iter` yields at src/lib.rs:365.
                          It is pinned against `split_regex` by tests/test_oracle.py.
* `byte_pair_merge`    -- src/lib.rs:140-196 (linear variant), including the leftmost
                          tie-break (`rank < min_rank.0`, strict) and the `i+3` span rule.
* `byte_pair_merge_large` -- src/lib.rs:47-138 (heap variant, ordered by (rank, start)).
* `byte_pair_encode`   -- src/lib.rs:198-211 (len 1 / <100 / heap dispatch).
* `encode_ordinary`    -- src/lib.rs:360-373 (whole-piece probe before merging).
* `encode`             -- src/lib.rs:375-442 (special-token loop; slice semantics).

The third-party regex arithmetic (fancy-regex 0.19 / regex 1.13, Cargo.toml:23-24) is
not under /root/reference; Python `regex` stands in for it, which is the substitution
the reference itself makes at core.py:395-404.
"""
from __future__ import annotations

import heapq

import regex

# --------------------------------------------------------------------------------------
# The three stock patterns (data; must equal tiktoken_ext/openai_public.py:12-14,89,104-114)
# --------------------------------------------------------------------------------------
R50K_PAT = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}++| ?\p{N}++| ?[^\s\p{L}\p{N}]++|\s++$|\s+(?!\S)|\s"""
CL100K_PAT = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}++|\p{N}{1,3}+| ?[^\s\p{L}\p{N}]++[\r\n]*+|\s++$|\s*[\r\n]|\s+(?!\S)|\s"""
O200K_PAT = "|".join(
    [
        r"""[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?""",
        r"""[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?""",
        r"""\p{N}{1,3}""",
        r""" ?[^\s\p{L}\p{N}]+[\r\n/]*""",
        r"""\s*[\r\n]+""",
        r"""\s+(?!\S)""",
        r"""\s+""",
    ]
)
# The original GPT-2 spelling the reference declares equivalent to R50K_PAT
# (openai_public.py:9-11).
GPT2_ORIG_PAT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?[\p{L}]+| ?[\p{N}]+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""

PAT_R50K, PAT_CL100K, PAT_O200K = 0, 1, 2
PATTERNS = {PAT_R50K: R50K_PAT, PAT_CL100K: CL100K_PAT, PAT_O200K: O200K_PAT}

# Class codes -- same numbering as tools/gen_unicode_tables.py
CONT, NL, SP, WSO, LU, LL, LC, MK, NU, AP, SL, OT, END = range(13)

_cls_cache: dict[str, int] = {}
_re_ws = regex.compile(r"\s")
_re_l = regex.compile(r"\p{L}")
_re_lu = regex.compile(r"[\p{Lu}\p{Lt}]")
_re_ll = regex.compile(r"\p{Ll}")
_re_m = regex.compile(r"\p{M}")
_re_n = regex.compile(r"\p{N}")


def char_class(ch: str) -> int:
    c = _cls_cache.get(ch)
    if c is not None:
        return c
    if _re_ws.match(ch):
        c = NL if ch in "\r\n" else (SP if ch == " " else WSO)
    elif _re_l.match(ch):
        c = LU if _re_lu.match(ch) else (LL if _re_ll.match(ch) else LC)
    elif _re_m.match(ch):
        c = MK
    elif _re_n.match(ch):
        c = NU
    elif ch == "'":
        c = AP
    elif ch == "/":
        c = SL
    else:
        c = OT
    _cls_cache[ch] = c
    return c


_compiled: dict[str, "regex.Pattern"] = {}


def split_regex(pat_str: str, text: str) -> list[str]:
    """Pieces exactly as `regex.findall(pat_str, text)` (core.py:395-404)."""
    p = _compiled.get(pat_str)
    if p is None:
        p = _compiled[pat_str] = regex.compile(pat_str)
    return p.findall(text)


# --------------------------------------------------------------------------------------
# Sequential scanner restatement
# --------------------------------------------------------------------------------------
def _is_ws(c):
    return c in (NL, SP, WSO)


def _is_letter(c):  # \p{L}
    return c in (LU, LL, LC)


def _contraction_len(text: str, p: int, end: int, ci: bool) -> int:
    """Length (in chars, incl. the apostrophe) of a contraction at text[p]=="'" or 0.

    r50k: '(?:[sdmt]|ll|ve|re) case-sensitive (openai_public.py:13);
    cl100k/o200k: (?i:...) -- Unicode simple case folding, so U+017F (long s) matches `s`.
    """
    if p + 1 >= end:
        return 0
    a = text[p + 1]
    if ci:
        if a in "sSſdDmMtT":
            return 2
        if p + 2 < end:
            b = text[p + 2]
            if (a in "lL" and b in "lL") or (a in "vV" and b in "eE") or (a in "rR" and b in "eE"):
                return 3
        return 0
    if a in "sdmt":
        return 2
    if p + 2 < end:
        b = text[p + 2]
        if (a == "l" and b == "l") or (a == "v" and b == "e") or (a == "r" and b == "e"):
            return 3
    return 0


def _ws_tail(cls, p, end, pat):
    """Whitespace alternatives: \\s++$ (r50k, cl100k), \\s*[\\r\\n]+? (cl100k, o200k),
    \\s+(?!\\S), \\s / \\s+."""
    q = p
    last_nl = -1
    while q < end and _is_ws(cls[q]):
        if cls[q] == NL:
            last_nl = q
        q += 1
    if pat != PAT_O200K and q == end:
        return q
    if pat != PAT_R50K and last_nl >= 0:
        return last_nl + 1
    if q == end:
        return q
    if q - p >= 2:
        return q - 1
    return p + 1


def _o200k_word(text, cls, s, end):
    """Letter alternatives 1 and 2 of the o200k pattern from start `s` (after the optional
    one-char prefix).  Returns the match end or -1."""
    r_end = s
    last_c = -1
    while r_end < end and cls[r_end] in (LU, LC, MK):
        if cls[r_end] != LU:
            last_c = r_end
        r_end += 1
    t_end = r_end
    while t_end < end and cls[t_end] in (LL, LC, MK):
        t_end += 1
    if t_end > r_end:
        e = t_end
    elif last_c >= 0:
        e = last_c + 1  # alternative 1 after backtracking: [..]* gives back to the last caseless char
    elif r_end > s:
        e = r_end  # alternative 2
    else:
        return -1
    if e < end and cls[e] == AP:
        e += _contraction_len(text, e, end, True)
    return e


def scan_piece_end(text: str, cls: list[int], p: int, end: int, pat: int) -> int:
    """End (exclusive) of the piece that starts at p, assuming p is a piece start."""
    c = cls[p]
    nxt = cls[p + 1] if p + 1 < end else END
    if pat == PAT_R50K:
        if c == AP:
            n = _contraction_len(text, p, end, False)
            if n:
                return p + n
        s = p + 1 if (c == SP and p + 1 < end) else p
        k = cls[s]
        if _is_letter(k):
            e = s
            while e < end and _is_letter(cls[e]):
                e += 1
            return e
        if k == NU:
            e = s
            while e < end and cls[e] == NU:
                e += 1
            return e
        if k in (MK, AP, SL, OT):
            e = s
            while e < end and cls[e] in (MK, AP, SL, OT):
                e += 1
            return e
        return _ws_tail(cls, p, end, pat)
    if pat == PAT_CL100K:
        if c == AP:
            n = _contraction_len(text, p, end, True)
            if n:
                return p + n
        if _is_letter(c) or (c not in (NL, NU) and _is_letter(nxt)):
            e = p + 1
            while e < end and _is_letter(cls[e]):
                e += 1
            return e
        if c == NU:
            e = p
            while e < end and e < p + 3 and cls[e] == NU:
                e += 1
            return e
        s = p + 1 if (c == SP and p + 1 < end) else p
        if cls[s] in (MK, AP, SL, OT):
            e = s
            while e < end and cls[e] in (MK, AP, SL, OT):
                e += 1
            while e < end and cls[e] == NL:
                e += 1
            return e
        return _ws_tail(cls, p, end, pat)
    # o200k
    if c in (LU, LL, LC, MK):
        return _o200k_word(text, cls, p, end)
    if c not in (NL, NU) and nxt in (LU, LL, LC, MK):
        return _o200k_word(text, cls, p + 1, end)
    if c == NU:
        e = p
        while e < end and e < p + 3 and cls[e] == NU:
            e += 1
        return e
    s = p + 1 if (c == SP and p + 1 < end) else p
    if cls[s] in (MK, AP, SL, OT):
        e = s
        while e < end and cls[e] in (MK, AP, SL, OT):
            e += 1
        while e < end and cls[e] in (NL, SL):
            e += 1
        return e
    return _ws_tail(cls, p, end, pat)


def split_scan(pat: int, text: str) -> list[str]:
    cls = [char_class(ch) for ch in text]
    out = []
    p, end = 0, len(text)
    while p < end:
        e = scan_piece_end(text, cls, p, end, pat)
        assert e > p
        out.append(text[p:e])
        p = e
    return out


# --------------------------------------------------------------------------------------
# byte_pair_merge / encode
# --------------------------------------------------------------------------------------
RANK_MAX = 0xFFFFFFFF


def byte_pair_merge(ranks: dict[bytes, int], piece: bytes) -> list[int]:
    """src/lib.rs:140-196: returns the list of part start offsets (plus the end sentinel)."""
    n = len(piece)
    parts = []
    min_rank, min_i = RANK_MAX, -1
    for i in range(n - 1):
        r = ranks.get(piece[i:i + 2], RANK_MAX)
        if r < min_rank:
            min_rank, min_i = r, i
        parts.append([i, r])
    parts.append([n - 1, RANK_MAX])
    parts.append([n, RANK_MAX])

    def get_rank(i):
        if i + 3 < len(parts):
            return ranks.get(piece[parts[i][0]:parts[i + 3][0]], RANK_MAX)
        return RANK_MAX

    while min_rank != RANK_MAX:
        i = min_i
        if i > 0:
            parts[i - 1][1] = get_rank(i - 1)
        parts[i][1] = get_rank(i)
        del parts[i + 1]
        min_rank, min_i = RANK_MAX, -1
        for j in range(len(parts) - 1):
            if parts[j][1] < min_rank:
                min_rank, min_i = parts[j][1], j
    return [p[0] for p in parts]


def byte_pair_merge_large(ranks: dict[bytes, int], piece: bytes) -> list[int]:
    """src/lib.rs:47-138: heap of (rank, start) with lazy invalidation."""
    n = len(piece)
    prev = [0] * n
    endv = [0] * n
    next_end = [0] * n
    next_rank = [RANK_MAX] * n
    cur_rank = [RANK_MAX] * n
    prev[0], endv[0], next_end[0] = -1, 1, 2
    heap = []
    for i in range(n - 1):
        r = ranks.get(piece[i:i + 2])
        if r is not None:
            heap.append((r, i))
            next_rank[i] = r
        prev[i + 1], endv[i + 1], next_end[i + 1] = i, i + 2, i + 3
    heapq.heapify(heap)

    def potential_merge(start, next_end_item):
        next_end[start] = next_end_item
        next_rank[start] = RANK_MAX
        if next_end_item <= n:
            r = ranks.get(piece[start:next_end_item])
            if r is not None:
                heapq.heappush(heap, (r, start))
                next_rank[start] = r

    while heap:
        r, left = heapq.heappop(heap)
        if r == RANK_MAX:
            break
        if r != next_rank[left]:
            continue
        right_start = endv[left]
        right_end = next_end[left]
        right_next_end = next_end[right_start]
        cur_rank[left] = next_rank[left]
        endv[left] = right_end
        potential_merge(left, right_next_end)
        if right_end < n:
            prev[right_end] = left
        if left > 0:
            potential_merge(prev[left], right_end)
        next_rank[right_start] = RANK_MAX
    out = []
    i = 0
    while i < n:
        out.append(cur_rank[i] if cur_rank[i] != RANK_MAX else ranks[piece[i:endv[i]]])
        i = endv[i]
    return out


def byte_pair_encode(piece: bytes, ranks: dict[bytes, int]) -> list[int]:
    """src/lib.rs:198-211."""
    if len(piece) == 1:
        return [ranks[piece]]
    if len(piece) < 100:
        parts = byte_pair_merge(ranks, piece)
        return [ranks[piece[a:b]] for a, b in zip(parts[:-1], parts[1:])]
    return byte_pair_merge_large(ranks, piece)


def encode_single_piece(piece: bytes, ranks: dict[bytes, int]) -> list[int]:
    """src/py.rs:145-150 == the body of the loop at src/lib.rs:366-370."""
    t = ranks.get(piece)
    if t is not None:
        return [t]
    return byte_pair_encode(piece, ranks)


def encode_ordinary(text: str, pat_str: str, ranks: dict[bytes, int], splitter=None) -> list[int]:
    """src/lib.rs:360-373 with the regex split done by Python `regex` (core.py:395-404)."""
    pieces = splitter(text) if splitter else split_regex(pat_str, text)
    out: list[int] = []
    for piece in pieces:
        out.extend(encode_single_piece(piece.encode("utf-8"), ranks))
    return out


def find_special(text: str, specials: dict[str, int], start: int, allowed) -> tuple[int, str] | None:
    """Next allowed special at or after `start` (src/lib.rs:389-401).

    The reference builds an alternation of the escaped special strings in hash-map order
    (lib.rs:625-631), so when one special is a prefix of another its choice is unspecified;
    this oracle (and the HIP path) resolve it as "longest allowed special at the leftmost
    position", which is one of the behaviours the reference can exhibit.
    """
    best = None
    for s in specials:
        if s not in allowed:
            continue
        i = text.find(s, start)
        if i < 0:
            continue
        if best is None or i < best[0] or (i == best[0] and len(s) > len(best[1])):
            best = (i, s)
    return best


def encode(text: str, pat_str: str, ranks: dict[bytes, int], specials: dict[str, int],
           allowed_special, splitter=None) -> list[int]:
    """src/lib.rs:375-442: split at allowed specials; each slice is an independent haystack."""
    allowed = set(allowed_special) & set(specials)
    out: list[int] = []
    start = 0
    while True:
        hit = find_special(text, specials, start, allowed) if allowed else None
        end = hit[0] if hit else len(text)
        out.extend(encode_ordinary(text[start:end], pat_str, ranks, splitter))
        if hit is None:
            break
        out.append(specials[hit[1]])
        start = hit[0] + len(hit[1])
    return out


# ------------------------------------------------------------------------------------------
# Byte-level and unstable entry points (src/lib.rs:444-599, src/py.rs:72-131), restated statement by statement.
# Pure Python over the functions above; used by the tests to check CoreBPE._encode_bytes / encode_with_unstable.
# ------------------------------------------------------------------------------------------
def encode_with_last_len(text: str, pat_str: str, ranks: dict[bytes, int], specials: dict[str, int],
                         allowed_special, splitter=None) -> tuple[list[int], int]:
    """src/lib.rs:375-442 including its second result, `last_piece_token_len`: the number of tokens that came
    from the last regex piece (0 right after a special token, lib.rs:433)."""
    allowed = set(allowed_special) & set(specials)
    out: list[int] = []
    start, last = 0, 0
    while True:
        hit = find_special(text, specials, start, allowed) if allowed else None
        end = hit[0] if hit else len(text)
        sl = text[start:end]
        for piece in (splitter(sl) if splitter else split_regex(pat_str, sl)):
            toks = encode_single_piece(piece.encode("utf-8"), ranks)
            last = len(toks)
            out.extend(toks)
        if hit is None:
            break
        out.append(specials[hit[1]])
        start = hit[0] + len(hit[1])
        last = 0
    return out, last


def increase_last_piece_token_len(tokens: list[int], last: int, decoder: dict[int, bytes]) -> int:
    """src/lib.rs:444-481."""
    def all_space(tok: int) -> bool:
        b = decoder.get(tok)
        return b is not None and all(c in b" \n\t" for c in b)

    if last > 0 and all_space(tokens[len(tokens) - last]):
        while last < len(tokens) and all_space(tokens[len(tokens) - last - 1]):
            last += 1
    assert last <= len(tokens)
    return last


# char::is_whitespace (Rust) = the Unicode White_Space property: 25 code points.  (Python's str.isspace() also accepts
# U+001C..U+001F, which Rust does not.)
RUST_WHITE_SPACE = frozenset(map(chr, [9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F,
                                       0x205F, 0x3000]))


def _decode_last_utf8(b: bytes) -> tuple[str | None, int]:
    """bstr::decode_last_utf8 (bstr 1.x) as lib.rs:581 uses it: the last code point of `b` and its byte length;
    (None, n) with n >= 1 when the tail is not valid UTF-8 (n = length of the maximal invalid suffix bstr reports:
    at most 3 trailing bytes that do not form a complete scalar)."""
    if not b:
        return None, 0
    start = len(b) - 1
    limit = max(0, len(b) - 4)
    while start > limit and (b[start] & 0xC0) == 0x80:
        start -= 1
    for s in range(start, len(b)):
        try:
            ch = b[s:].decode("utf-8")
        except UnicodeDecodeError:
            continue
        if len(ch) == 1:
            return ch, len(b) - s
    return None, 1


def encode_unstable_native(text: str, pat_str: str, ranks: dict[bytes, int], specials: dict[str, int],
                           allowed_special, splitter=None) -> tuple[list[int], set[tuple[int, ...]]]:
    """src/lib.rs:483-599."""
    decoder = {v: k for k, v in ranks.items()}
    tokens, last = encode_with_last_len(text, pat_str, ranks, specials, allowed_special, splitter)
    if last == 0:
        return tokens, set()
    last = increase_last_piece_token_len(tokens, last, decoder)
    unstable = b"".join(decoder[t] for t in tokens[len(tokens) - last:])
    del tokens[len(tokens) - last:]
    completions: set[tuple[int, ...]] = set()
    if not unstable:
        return tokens, completions
    import bisect

    sorted_tokens = sorted(ranks)
    point = bisect.bisect_left(sorted_tokens, unstable)
    while point < len(sorted_tokens) and sorted_tokens[point].startswith(unstable):
        completions.add((ranks[sorted_tokens[point]],))
        point += 1
    for i in range(1, len(unstable)):
        prefix, suffix = unstable[:i], unstable[i:]
        point = bisect.bisect_left(sorted_tokens, suffix)
        while point < len(sorted_tokens) and sorted_tokens[point].startswith(suffix):
            possibility = prefix + sorted_tokens[point]
            try:
                encoded = encode_ordinary(possibility.decode("utf-8"), pat_str, ranks, splitter)
            except UnicodeDecodeError:
                encoded = byte_pair_encode(possibility, ranks)
            seq, seq_len = [], 0
            for t in encoded:
                seq.append(t)
                seq_len += len(decoder[t])
                if seq_len >= len(unstable):
                    break
            completions.add(tuple(seq))
            point += 1
    if len(unstable) > 1:
        ch, k = _decode_last_utf8(unstable)
        if len(unstable) - k > 0 and ch is not None and ch in RUST_WHITE_SPACE:
            re_enc = byte_pair_encode(unstable[:len(unstable) - k], ranks) + byte_pair_encode(unstable[len(unstable) - k:], ranks)
            completions.add(tuple(re_enc))
    return tokens, completions


def encode_bytes(data: bytes, pat_str: str, ranks: dict[bytes, int], splitter=None) -> list[int]:
    """src/py.rs:72-115."""
    try:
        return encode_ordinary(data.decode("utf-8"), pat_str, ranks, splitter)
    except UnicodeDecodeError as e:
        valid = e.start
    decoder = {v: k for k, v in ranks.items()}
    tokens, last = encode_with_last_len(data[:valid].decode("utf-8"), pat_str, ranks, {}, set(), splitter)
    last = increase_last_piece_token_len(tokens, last, decoder)
    if tokens and last > 0:
        unstable = b"".join(decoder[t] for t in tokens[len(tokens) - last:]) + data[valid:]
        del tokens[len(tokens) - last:]
    else:
        unstable = data[valid:]
    if unstable:
        t = ranks.get(unstable)
        tokens.extend([t] if t is not None else byte_pair_encode(unstable, ranks))
    return tokens
