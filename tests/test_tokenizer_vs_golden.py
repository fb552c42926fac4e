"""CPU: wan2gp_amd.tokenizers.HuggingfaceTokenizer against tests/golden/tokenizer.json, produced by the reference's own class
(models/wan/modules/tokenizers.py:44-82) on the committed tiny Unigram tokenizer (oracle/make_golden_tokenizer.py): cleaned
strings, ids and masks must be equal for every cleaning mode, with padding / truncation to seq_len."""
import json
import os

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
GOLD = json.load(open(os.path.join(G, "tokenizer.json")))


@pytest.mark.parametrize("clean", ["whitespace", "lower", "canonicalize", None])
def test_tokenizer_matches_the_reference(clean):
    from wan2gp_amd.tokenizers import HuggingfaceTokenizer
    t = HuggingfaceTokenizer(os.path.join(G, "tiny_tokenizer"), seq_len=16, clean=clean)
    case = GOLD["cases"][str(clean)]
    if clean:
        assert [t._clean(p) for p in GOLD["prompts"]] == case["cleaned"]
    ids, mask = t(GOLD["prompts"], return_mask=True, add_special_tokens=True)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (len(GOLD["prompts"]), 16)
    assert ids.tolist() == case["ids"] and mask.tolist() == case["mask"]
    assert t.vocab_size == GOLD["vocab_size"]


def test_unpadded_single_string_and_non_ascii_cleanup():
    from wan2gp_amd import tokenizers as T
    t = T.HuggingfaceTokenizer(os.path.join(G, "tiny_tokenizer"), seq_len=None, clean="whitespace")
    assert t(GOLD["prompts"][0]).tolist() == GOLD["cases"]["unpadded_single"]["ids"]
    # the ftfy-free subset (documented as unpinned): curly quotes, ligatures, full-width forms, double-escaped entities
    assert T.whitespace_clean(T.basic_clean("  a “cat”  &amp;amp; dog\r\nﬁne ＡＢ ")) == 'a "cat" & dog fine AB'
