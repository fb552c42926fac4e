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


def basic_clean(text):
    text = _ftfy.fix_text(text) if _ftfy is not None else _fix_text_subset(text)
    text = html.unescape(html.unescape(text))
    return text.strip()


def whitespace_clean(text):
    text = re.sub(r"\s+", " ", text)
    return text.strip()


def canonicalize(text, keep_punctuation_exact_string=None):
    text = text.replace("_", " ")
    if keep_punctuation_exact_string:
        text = keep_punctuation_exact_string.join(part.translate(str.maketrans("", "", string.punctuation))
                                                  for part in text.split(keep_punctuation_exact_string))
    else:
        text = text.translate(str.maketrans("", "", string.punctuation))
    text = text.lower()
    text = re.sub(r"\s+", " ", text)
    return text.strip()


class HuggingfaceTokenizer:
    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        assert clean in (None, "whitespace", "lower", "canonicalize")
        self.name, self.seq_len, self.clean = name, seq_len, clean
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def __call__(self, sequence, **kwargs):
        return_mask = kwargs.pop("return_mask", False)
        _kwargs = {"return_tensors": "pt"}
        if self.seq_len is not None:
            _kwargs.update({"padding": "max_length", "truncation": True, "max_length": self.seq_len})
        _kwargs.update(**kwargs)
        if isinstance(sequence, str):
            sequence = [sequence]
        if self.clean:
            sequence = [self._clean(u) for u in sequence]
        ids = self.tokenizer(sequence, **_kwargs)
        if return_mask:
            return ids.input_ids, ids.attention_mask
        return ids.input_ids

    def _clean(self, text):
        if self.clean == "whitespace":
            text = whitespace_clean(basic_clean(text))
        elif self.clean == "lower":
            text = whitespace_clean(basic_clean(text)).lower()
        elif self.clean == "canonicalize":
            text = canonicalize(basic_clean(text))
        return text
