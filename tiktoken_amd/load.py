"""Vocabulary file loading (reference tiktoken/load.py): `.tiktoken` files (`base64(token) SP rank`
per line), the GPT-2 `vocab.bpe` + `encoder.json` pair, and the sha256-checked download cache.

Out of the accelerated path (one-time I/O); kept so that the stock constructors of
tiktoken_ext.openai_public work whenever the pinned files are reachable or already cached.
"""
from __future__ import annotations

import base64
import gzip
import hashlib
import json
import os
import tempfile
import uuid


def read_file(blobpath: str) -> bytes:
    if "://" not in blobpath:
        with open(blobpath, "rb", buffering=0) as f:
            return f.read()
    if blobpath.startswith(("http://", "https://")):
        import requests  # plain HTTP(S): no blobfile needed for the public files

        resp = requests.get(blobpath)
        resp.raise_for_status()
        return resp.content
    try:
        import blobfile
    except ImportError as e:
        raise ImportError("blobfile is not installed. Please install it by running `pip install blobfile`.") from e
    return blobfile.read_bytes(blobpath)


def check_hash(data: bytes, expected_hash: str) -> bool:
    return hashlib.sha256(data).hexdigest() == expected_hash


def _cache_dir() -> tuple[str, bool]:
    for var in ("TIKTOKEN_CACHE_DIR", "DATA_GYM_CACHE_DIR"):
        if var in os.environ:
            return os.environ[var], True
    return os.path.join(tempfile.gettempdir(), "data-gym-cache"), False


def read_file_cached(blobpath: str, expected_hash: str | None = None) -> bytes:
    cache_dir, user_specified = _cache_dir()
    if cache_dir == "":  # caching disabled
        return read_file(blobpath)
    cache_path = os.path.join(cache_dir, hashlib.sha1(blobpath.encode()).hexdigest())
    if os.path.exists(cache_path):
        with open(cache_path, "rb", buffering=0) as f:
            data = f.read()
        if expected_hash is None or check_hash(data, expected_hash):
            return data
        try:  # stale / corrupt cache entry
            os.remove(cache_path)
        except OSError:
            pass
    contents = read_file(blobpath)
    if expected_hash and not check_hash(contents, expected_hash):
        raise ValueError(
            f"Hash mismatch for data downloaded from {blobpath} (expected {expected_hash}). "
            f"This may indicate a corrupted download. Please try again."
        )
    try:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = f"{cache_path}.{uuid.uuid4()}.tmp"
        with open(tmp, "wb") as f:
            f.write(contents)
        os.rename(tmp, cache_path)
    except OSError:
        if user_specified:  # only the implicit default cache may fail silently
            raise
    return contents


def data_gym_byte_order() -> list[int]:
    """Printable non-space bytes first, then the rest: the GPT-2 byte <-> rank convention."""
    order = [b for b in range(256) if chr(b).isprintable() and chr(b) != " "]
    return order + [b for b in range(256) if b not in order]


def data_gym_to_mergeable_bpe_ranks(vocab_bpe_file: str, encoder_json_file: str, vocab_bpe_hash: str | None = None,
                                    encoder_json_hash: str | None = None, clobber_one_byte_tokens: bool = False) -> dict[bytes, int]:
    rank_to_byte = data_gym_byte_order()
    n_printable = sum(1 for b in range(256) if chr(b).isprintable() and chr(b) != " ")
    char_to_byte = {chr(b): b for b in rank_to_byte[:n_printable]}
    for i, b in enumerate(rank_to_byte[n_printable:]):
        char_to_byte[chr(256 + i)] = b

    def to_bytes(s: str) -> bytes:
        return bytes(char_to_byte[ch] for ch in s)

    merges_text = read_file_cached(vocab_bpe_file, vocab_bpe_hash).decode()
    merges = [tuple(line.split()) for line in merges_text.split("\n")[1:-1]]
    ranks = {bytes([b]): i for i, b in enumerate(rank_to_byte)}
    for first, second in merges:
        ranks[to_bytes(first) + to_bytes(second)] = len(ranks)
    encoder = {to_bytes(k): v for k, v in json.loads(read_file_cached(encoder_json_file, encoder_json_hash)).items()}
    encoder.pop(b"<|endoftext|>", None)
    encoder.pop(b"<|startoftext|>", None)
    if clobber_one_byte_tokens:
        for k, v in encoder.items():
            if len(k) == 1:
                ranks[k] = v
    assert ranks == encoder  # merge order must equal token index order
    return ranks


def dump_tiktoken_bpe(bpe_ranks: dict[bytes, int], tiktoken_bpe_file: str) -> None:
    lines = [base64.b64encode(tok) + b" " + str(rank).encode() + b"\n" for tok, rank in sorted(bpe_ranks.items(), key=lambda kv: kv[1])]
    if "://" in tiktoken_bpe_file:
        try:
            import blobfile
        except ImportError as e:
            raise ImportError("blobfile is not installed. Please install it by running `pip install blobfile`.") from e
        with blobfile.BlobFile(tiktoken_bpe_file, "wb") as f:
            f.writelines(lines)
        return
    with open(tiktoken_bpe_file, "wb") as f:
        f.writelines(lines)


def parse_tiktoken_bpe(contents: bytes, source: str = "<bytes>") -> dict[bytes, int]:
    ranks: dict[bytes, int] = {}
    for line in contents.splitlines():
        if not line:
            continue
        try:
            token, rank = line.split()
            ranks[base64.b64decode(token)] = int(rank)
        except Exception as e:
            raise ValueError(f"Error parsing line {line!r} in {source}") from e
    return ranks


def load_tiktoken_bpe(tiktoken_bpe_file: str, expected_hash: str | None = None) -> dict[bytes, int]:
    contents = read_file_cached(tiktoken_bpe_file, expected_hash)
    if tiktoken_bpe_file.endswith(".gz"):
        contents = gzip.decompress(contents)
    return parse_tiktoken_bpe(contents, tiktoken_bpe_file)
