"""Prompt tokenizer in front of the UMT5 encoder -- `HuggingfaceTokenizer` of models/wan/modules/tokenizers.py:44-82
("UMT5-XXL text encoder (+ tokenizer)").  Host string work: the sentencepiece model itself is
the `transformers` AutoTokenizer the reference uses; what is restated here is the reference's cleaning + padding contract:

    HuggingfaceTokenizer(name, seq_len=512, clean='whitespace')(texts, return_mask=True, add_special_tokens=True)
        -> (input_ids [B, seq_len] int64, attention_mask [B, seq_len] int64), padded to seq_len, truncated

`basic_clean` is ftfy.fix_text + two html.unescape passes in the reference.  ftfy is used when it is installed; without it the
subset of its default fixes that changes ordinary prompts is applied (NFC, curly quotes, Latin ligatures, full-width forms,
control characters, line breaks) -- for plain ASCII text both are the identity.  **Parity for non-ASCII clean-up is unpinned
where ftfy is absent** (it is absent from the build container; tests/golden/tokenizer.json covers ASCII + HTML entities).
"""
import html
import string
import unicodedata

import regex as re

__all__ = ["HuggingfaceTokenizer"]

try:                                                   # tokenizers.py:5
    import ftfy as _ftfy
except ImportError:                                    # pragma: no cover - depends on the environment
    _ftfy = None

_QUOTES = {0x2018: "'", 0x2019: "'", 0x201a: "'", 0x201b: "'", 0x201c: '"', 0x201d: '"', 0x201e: '"', 0x201f: '"',
           0xff02: '"', 0xff07: "'"}
_LIGATURES = {0xfb00: "ff", 0xfb01: "fi", 0xfb02: "fl", 0xfb03: "ffi", 0xfb04: "ffl", 0xfb05: "ſt", 0xfb06: "st"}
_CONTROL = {c: None for c in list(range(0x00, 0x09)) + [0x0b] + list(range(0x0e, 0x20)) + [0x7f, 0xfeff]}


def _fix_text_subset(text):
    text = text.translate(_QUOTES).translate(_LIGATURES)
    text = "".join(unicodedata.normalize("NFKC", ch) if 0xff01 <= ord(ch) <= 0xff5e else ch for ch in text)   # full-width ASCII
    text = text.replace("\r\n", "\n").replace("\r", "\n").replace("\u2028", "\n").replace("\u2029", "\n").replace("\x85", "\n")
    text = text.translate(_CONTROL)
    return unicodedata.normalize("NFC", text)


_WS = re.compile(r"\s+")
_NO_PUNCT = str.maketrans("", "", string.punctuation)


def _unescape_and_fix(prompt):
    """Mojibake / entity repair: ftfy when present, otherwise its prompt-relevant subset; HTML entities decoded twice."""
    repaired = _ftfy.fix_text(prompt) if _ftfy is not None else _fix_text_subset(prompt)
    for _ in range(2):
        repaired = html.unescape(repaired)
    return repaired.strip()


def _squeeze(prompt):
    return _WS.sub(" ", prompt).strip()


def _canonical(prompt, keep=None):
    """Underscores to spaces, punctuation dropped (except the exact string `keep`), lower case, single spaces."""
    prompt = prompt.replace("_", " ")
    pieces = prompt.split(keep) if keep else [prompt]
    prompt = (keep or "").join(piece.translate(_NO_PUNCT) for piece in pieces)
    return _squeeze(prompt.lower())


# cleaning mode -> function of the raw prompt (the reference's three `clean` settings)
_CLEANERS = {
    None: lambda prompt: prompt,
    "whitespace": lambda prompt: _squeeze(_unescape_and_fix(prompt)),
    "lower": lambda prompt: _squeeze(_unescape_and_fix(prompt)).lower(),
    "canonicalize": lambda prompt: _canonical(_unescape_and_fix(prompt)),
}
basic_clean, whitespace_clean, canonicalize = _unescape_and_fix, _squeeze, _canonical      # the reference module's function names


class HuggingfaceTokenizer:
    """Wraps `transformers.AutoTokenizer.from_pretrained(name)`: prompts are cleaned by mode, then padded / truncated to `seq_len`."""

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        if clean not in _CLEANERS:
            raise AssertionError(f"clean must be one of {sorted(k for k in _CLEANERS if k)} or None, got {clean!r}")
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def _encode_kwargs(self, overrides):
        kw = {"return_tensors": "pt"}
        if self.seq_len is not None:
            kw.update(padding="max_length", truncation=True, max_length=self.seq_len)
        kw.update(overrides)
        return kw

    def __call__(self, sequence, **kwargs):
        want_mask = kwargs.pop("return_mask", False)
        prompts = [sequence] if isinstance(sequence, str) else list(sequence)
        cleaner = _CLEANERS[self.clean]
        enc = self.tokenizer([cleaner(p) for p in prompts], **self._encode_kwargs(kwargs))
        return (enc.input_ids, enc.attention_mask) if want_mask else enc.input_ids

    def _clean(self, text):
        return _CLEANERS[self.clean](text)
