"""`Encoding` -- the user-facing tokenizer object, call-for-call compatible with the reference's
`tiktoken.core.Encoding` (reference tiktoken/core.py:16-428), on top of the HIP `CoreBPE`.

Differences that matter for speed, none for results:
  * `encode_ordinary_batch` / `encode_batch` hand the WHOLE batch to the GPU in one call
    (`CoreBPE.encode_batch_packed`) instead of mapping documents over a thread pool
    (core.py:174-176, 202-206); `num_threads` is accepted for compatibility.
  * `encode_*_batch_packed` / `encode_to_numpy` return numpy arrays and skip the
    `list[list[int]]` materialisation, which costs more than the encode itself.
"""
from __future__ import annotations

import functools
from typing import TYPE_CHECKING, AbstractSet, Collection, Literal, NoReturn, Sequence

import numpy as np

from . import _tiktoken

if TYPE_CHECKING:
    import re

    import numpy.typing as npt

_SURROGATE_FIX = ("utf-16", "surrogatepass", "utf-16", "replace")


def _repair_surrogates(text: str) -> str:
    # same repair as core.py:79,135: join surrogate pairs, replace lone ones with U+FFFD
    return text.encode(_SURROGATE_FIX[0], _SURROGATE_FIX[1]).decode(_SURROGATE_FIX[2], _SURROGATE_FIX[3])


def _utf8(text: str) -> bytes:
    try:
        return text.encode("utf-8")
    except UnicodeEncodeError:
        return _repair_surrogates(text).encode("utf-8")


class Encoding:
    def __init__(self, name: str, *, pat_str: str, mergeable_ranks: dict[bytes, int], special_tokens: dict[str, int],
                 explicit_n_vocab: int | None = None):
        """See tiktoken_ext/openai_public.py for how the stock encodings call this.

        name: identifies the behaviour (encodings with different special tokens need different names).
        pat_str: the regex that splits text into pieces before BPE, compiled once (reference: src/lib.rs:623).  The r50k/gpt2, cl100k and
            o200k patterns -- in any of their spellings, with variations of the contraction list, the digit group length, the suffix set
            after punctuation and the white-space rules, e.g. Qwen2's or Llama-3's -- run on hand-written GPU scanners; any other
            pattern runs on the generic GPU regex engine (classes, \\p{..} General_Category values and scripts, groups, (?i: ), greedy / lazy /
            possessive quantifiers, atomic groups, look-ahead, \\b, look-behind of fixed length).  ValueError with the reason for what neither takes:
            look-behind of variable length, back-references, binary properties, patterns that can match the empty string.  There is no CPU regex fallback.
        mergeable_ranks: token bytes -> rank; ranks are merge priorities.
        special_tokens: special token string -> id.
        explicit_n_vocab: if given, checked against the number of tokens and the largest id.
        """
        self.name = name
        self._pat_str = pat_str
        self._ranks = mergeable_ranks
        self._special_tokens = special_tokens
        self._special_token_values = set(special_tokens.values())
        # The reference checks explicit_n_vocab BEFORE it builds the core (core.py:96-101): an inconsistent explicit_n_vocab is an AssertionError
        # whatever else is wrong with the vocabulary, and nothing has touched the GPU by then.  A lazily parsed file (vocab_io.RankTable) gives
        # count and largest rank from its packed arrays without a dict walk; they are the dict's unless the file lists a token twice (the
        # dict keeps the later rank, as the reference's load.py:159-171 does).  So: with explicit_n_vocab and arrays not yet known to be
        # distinct, the DICT decides, before the core is built (the stock encodings that pass explicit_n_vocab have 50 k tokens: 40 ms);
        # without explicit_n_vocab nothing is asserted and the arrays' figures are corrected after tk_create if it met such a token.
        def check(n_tokens: int, top: int) -> None:
            self.max_token_value = max(top, max(special_tokens.values(), default=0))
            if explicit_n_vocab:
                assert n_tokens + len(special_tokens) == explicit_n_vocab
                assert self.max_token_value == explicit_n_vocab - 1

        pending = getattr(mergeable_ranks, "_pending", None)
        n_seen = None
        if pending is not None and not explicit_n_vocab and not getattr(mergeable_ranks, "_distinct", False):
            ids = pending[2]
            n_seen = len(ids)
            check(n_seen, int(ids.max()) if len(ids) else 0)
        else:
            check(len(mergeable_ranks), mergeable_ranks.max_rank() if hasattr(mergeable_ranks, "max_rank") else max(mergeable_ranks.values()))
        self._core_bpe = _tiktoken.CoreBPE(mergeable_ranks, special_tokens, pat_str)
        if n_seen is not None and len(mergeable_ranks) != n_seen:  # (a token listed twice: collapsed by now; nothing to assert, the figure follows the dict)
            check(len(mergeable_ranks), mergeable_ranks.max_rank())

    @property
    def _mergeable_ranks(self) -> dict[bytes, int]:
        """(a RankTable is filled before it leaves: C code that reads a dict's storage directly would see it empty otherwise)"""
        r = self._ranks
        return r.materialize() if hasattr(r, "materialize") else r

    @_mergeable_ranks.setter
    def _mergeable_ranks(self, ranks: dict[bytes, int]) -> None:
        # (the reference's attribute is a plain one: subclasses and patches assign it.  As there, the core that was built is not rebuilt.)
        self._ranks = ranks

    def __repr__(self) -> str:
        return f"<Encoding {self.name!r}>"

    # ------------------------------------------------------------------ special-token policy
    def _special_policy(self, allowed_special, disallowed_special):
        """Resolve the "all" shorthands (core.py:116-119)."""
        if allowed_special == "all":
            allowed_special = self.special_tokens_set
        if disallowed_special == "all":
            disallowed_special = self.special_tokens_set - allowed_special
        return allowed_special, disallowed_special

    @staticmethod
    def _reject_disallowed(text: str, disallowed_special) -> None:
        if not disallowed_special:
            return
        if not isinstance(disallowed_special, frozenset):
            disallowed_special = frozenset(disallowed_special)
        hit = _special_token_regex(disallowed_special).search(text)
        if hit:
            raise_disallowed_special_token(hit.group())

    # ------------------------------------------------------------------ encoding
    def encode_ordinary(self, text: str) -> list[int]:
        """Encode ignoring special tokens; same result as `encode(text, disallowed_special=())`."""
        try:
            return self._core_bpe.encode_ordinary(text)
        except UnicodeEncodeError:
            return self._core_bpe.encode_ordinary(_repair_surrogates(text))

    def encode(self, text: str, *, allowed_special: Literal["all"] | AbstractSet[str] = set(),  # noqa: B006
               disallowed_special: Literal["all"] | Collection[str] = "all") -> list[int]:
        """Encode a string.  Text that spells a special token raises ValueError unless the token is in
        `allowed_special` (then it is emitted as the special id) or removed from `disallowed_special`
        (then it is encoded as ordinary text).  Both arguments accept "all"."""
        allowed_special, disallowed_special = self._special_policy(allowed_special, disallowed_special)
        self._reject_disallowed(text, disallowed_special)
        try:
            return self._core_bpe.encode(text, allowed_special)
        except UnicodeEncodeError:
            return self._core_bpe.encode(_repair_surrogates(text), allowed_special)

    def encode_to_numpy(self, text: str, *, allowed_special: Literal["all"] | AbstractSet[str] = set(),  # noqa: B006
                        disallowed_special: Literal["all"] | Collection[str] = "all") -> "npt.NDArray[np.uint32]":
        """Like `encode`, returning a uint32 array without building a Python list."""
        allowed_special, disallowed_special = self._special_policy(allowed_special, disallowed_special)
        self._reject_disallowed(text, disallowed_special)
        buffer = self._core_bpe.encode_to_tiktoken_buffer(text, allowed_special)
        return np.frombuffer(buffer, dtype=np.uint32)

    @staticmethod
    def _pack(texts: Sequence[str]):
        chunks = [_utf8(t) for t in texts]
        off = np.zeros(len(chunks) + 1, dtype=np.uint64)
        if chunks:
            np.cumsum(np.fromiter((len(c) for c in chunks), dtype=np.uint64, count=len(chunks)), out=off[1:])
        blob = np.frombuffer(b"".join(chunks), dtype=np.uint8)
        return blob, off

    @staticmethod
    def _unpack(tokens: np.ndarray, tok_off: np.ndarray) -> list[list[int]]:
        # one tolist per document: the ints are created once, straight into their list (a flat list sliced afterwards costs twice)
        bounds = tok_off.tolist()
        return [tokens[a:b].tolist() for a, b in zip(bounds[:-1], bounds[1:])]

    def encode_ordinary_batch_packed(self, text: Sequence[str]):
        """(tokens uint32[T], tok_off uint64[n+1]) for a batch, ignoring special tokens."""
        blob, off = self._pack(text)
        return self._core_bpe.encode_batch_packed(blob, off, None)

    def encode_batch_packed(self, text: Sequence[str], *, allowed_special: Literal["all"] | AbstractSet[str] = set(),  # noqa: B006
                            disallowed_special: Literal["all"] | Collection[str] = "all"):
        allowed_special, disallowed_special = self._special_policy(allowed_special, disallowed_special)
        if disallowed_special:
            for t in text:
                self._reject_disallowed(t, disallowed_special)
        blob, off = self._pack(text)
        return self._core_bpe.encode_batch_packed(blob, off, allowed_special)

    def encode_ordinary_batch(self, text: list[str], *, num_threads: int = 8) -> list[list[int]]:
        """Encode a list of strings, ignoring special tokens (one GPU batch; `num_threads` is kept for
        signature compatibility)."""
        return self._unpack(*self.encode_ordinary_batch_packed(text))

    def encode_batch(self, text: list[str], *, num_threads: int = 8,
                     allowed_special: Literal["all"] | AbstractSet[str] = set(),  # noqa: B006
                     disallowed_special: Literal["all"] | Collection[str] = "all") -> list[list[int]]:
        """Encode a list of strings; see `encode` for the special-token arguments."""
        return self._unpack(*self.encode_batch_packed(text, allowed_special=allowed_special,
                                                      disallowed_special=disallowed_special))

    def encode_with_unstable(self, text: str, *, allowed_special: Literal["all"] | AbstractSet[str] = set(),  # noqa: B006
                             disallowed_special: Literal["all"] | Collection[str] = "all") -> tuple[list[int], list[list[int]]]:
        """Stable prefix tokens plus the possible token sequences that could complete the unstable tail."""
        allowed_special, disallowed_special = self._special_policy(allowed_special, disallowed_special)
        self._reject_disallowed(text, disallowed_special)
        return self._core_bpe.encode_with_unstable(text, allowed_special)

    def encode_single_token(self, text_or_bytes: str | bytes) -> int:
        """Id of the token whose bytes are exactly the argument (special tokens included); KeyError otherwise."""
        if isinstance(text_or_bytes, str):
            text_or_bytes = text_or_bytes.encode("utf-8")
        return self._core_bpe.encode_single_token(text_or_bytes)

    # ------------------------------------------------------------------ decoding
    def decode_bytes(self, tokens: Sequence[int]) -> bytes:
        return self._core_bpe.decode_bytes(tokens)

    def decode(self, tokens: Sequence[int], errors: str = "replace") -> str:
        """Decode to str.  Lossy by default: token boundaries need not be UTF-8 boundaries, so invalid
        sequences are replaced unless `errors="strict"`."""
        return self._core_bpe.decode_bytes(tokens).decode("utf-8", errors=errors)

    def decode_single_token_bytes(self, token: int) -> bytes:
        return self._core_bpe.decode_single_token_bytes(token)

    def decode_tokens_bytes(self, tokens: Sequence[int]) -> list[bytes]:
        return [self.decode_single_token_bytes(t) for t in tokens]

    def decode_with_offsets(self, tokens: Sequence[int]) -> tuple[str, list[int]]:
        """Text plus, per token, the index of the character in which the token starts (a token that begins
        with a continuation byte is attributed to the character it continues).  Strict UTF-8."""
        pieces = self.decode_tokens_bytes(tokens)
        offsets: list[int] = []
        n_chars = 0
        for piece in pieces:
            starts_mid_char = 0x80 <= piece[0] < 0xC0
            offsets.append(max(0, n_chars - (1 if starts_mid_char else 0)))
            n_chars += sum(1 for b in piece if not 0x80 <= b < 0xC0)
        return b"".join(pieces).decode("utf-8", errors="strict"), offsets

    def decode_batch(self, batch: Sequence[Sequence[int]], *, errors: str = "replace", num_threads: int = 8) -> list[str]:
        """Decode a batch; the whole batch goes to the GPU in one call (`num_threads` is accepted for compatibility)."""
        data, bounds = self._decode_packed(batch)
        if data is None:
            return [self.decode_bytes(t).decode("utf-8", errors=errors) for t in batch]
        view = memoryview(data)  # (str() decodes a slice of the result buffer in place: no bytes object per document in between)
        return [str(view[a:b], "utf-8", errors) if b > a else "" for a, b in zip(bounds[:-1], bounds[1:])]

    def _decode_packed(self, batch: Sequence[Sequence[int]]):
        """(uint8 array of all bytes back to back -- a view of the library's result buffer --, list of n + 1 byte offsets), or (None, None)
        when the ids are too sparse for the device table."""
        import numpy as np

        lens = np.fromiter((len(t) for t in batch), dtype=np.uint64, count=len(batch))
        tok_off = np.zeros(len(batch) + 1, dtype=np.uint64)
        np.cumsum(lens, out=tok_off[1:])
        flat = np.fromiter((t for doc in batch for t in doc), dtype=np.uint32, count=int(tok_off[-1]))
        try:
            data, byte_off = self._core_bpe.decode_batch_packed(flat, tok_off, as_array=True)
        except ValueError:  # (ids too sparse for the device table)
            return None, None
        return data, byte_off.tolist()

    def decode_bytes_batch(self, batch: Sequence[Sequence[int]], *, num_threads: int = 8) -> list[bytes]:
        """One GPU call for the whole batch (tk_decode_batch) instead of one pool task per document."""
        data, bounds = self._decode_packed(batch)
        if data is None:
            return [self.decode_bytes(t) for t in batch]
        view = memoryview(data)
        return [bytes(view[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]

    def token_byte_values(self) -> list[bytes]:
        return self._core_bpe.token_byte_values()

    @property
    def eot_token(self) -> int:
        return self._special_tokens["<|endoftext|>"]

    @functools.cached_property
    def special_tokens_set(self) -> set[str]:
        return set(self._special_tokens.keys())

    def is_special_token(self, token: int) -> bool:
        assert isinstance(token, int)
        return token in self._special_token_values

    @property
    def n_vocab(self) -> int:
        """Kept for backwards compatibility; `max_token_value + 1`."""
        return self.max_token_value + 1

    # ------------------------------------------------------------------ private
    def _encode_single_piece(self, text_or_bytes: str | bytes) -> list[int]:
        """BPE of the argument's bytes with no regex split and no special tokens."""
        if isinstance(text_or_bytes, str):
            text_or_bytes = text_or_bytes.encode("utf-8")
        return self._core_bpe.encode_single_piece(text_or_bytes)

    def _encode_only_native_bpe(self, text: str) -> list[int]:
        """Regex split in Python (`regex` module), BPE per piece on the device."""
        import regex

        out: list[int] = []
        for piece in regex.findall(regex.compile(self._pat_str), text):
            out.extend(self._core_bpe.encode_single_piece(piece.encode("utf-8")))
        return out

    def _encode_bytes(self, text: bytes) -> list[int]:
        return self._core_bpe._encode_bytes(text)

    def __getstate__(self) -> object:
        from . import plugins

        if plugins._CATALOGUE.registered(self):
            return self.name  # registered encodings pickle by name
        return {"name": self.name, "pat_str": self._pat_str, "mergeable_ranks": self._mergeable_ranks,
                "special_tokens": self._special_tokens}

    def __setstate__(self, value: object) -> None:
        from . import plugins

        if isinstance(value, str):
            self.__dict__ = plugins.get_encoding(value).__dict__
            return
        self.__init__(**value)


@functools.lru_cache(maxsize=128)
def _special_token_regex(tokens: frozenset[str]) -> "re.Pattern[str]":
    try:
        import regex as re
    except ImportError:
        import re
    return re.compile("(" + "|".join(re.escape(t) for t in tokens) + ")")


def raise_disallowed_special_token(token: str) -> NoReturn:
    raise ValueError(
        f"Encountered text corresponding to disallowed special token {token!r}.\n"
        "If you want this text to be encoded as a special token, "
        f"pass it to `allowed_special`, e.g. `allowed_special={{{token!r}, ...}}`.\n"
        f"If you want this text to be encoded as normal text, disable the check for this token "
        f"by passing `disallowed_special=(enc.special_tokens_set - {{{token!r}}})`.\n"
        "To disable this check for all special tokens, pass `disallowed_special=()`.\n"
    )
