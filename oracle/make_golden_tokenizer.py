"""TEST INFRASTRUCTURE ONLY -- tests/golden/tokenizer.json (+ tests/golden/tiny_tokenizer/) from the REFERENCE's own
`HuggingfaceTokenizer` (models/wan/modules/tokenizers.py), executed in the build container:   python oracle/make_golden_tokenizer.py

The umt5-xxl sentencepiece files are not reachable (no network), so a tiny Unigram tokenizer with T5's special-token layout
(<pad> 0, </s> 1, <unk> 2) is trained on a fixed corpus and committed; the reference class and this repo's class both load it
through transformers.AutoTokenizer.  The reference module imports `ftfy`, which is absent here: it is stubbed with the identity,
so the fixture only holds inputs on which ftfy.fix_text IS the identity (ASCII text, HTML entities, odd whitespace)."""
import json
import os
import sys
import types

import importlib.util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
TOK_DIR = os.path.join(ROOT, "tests", "golden", "tiny_tokenizer")

CORPUS = ["a cat walks on the beach at sunset", "two dogs play in the snow", "a red car drives through the city at night",
          "the camera pans slowly over a mountain lake", "cinematic lighting, high quality, 4k", "a woman smiles & waves",
          "slow motion shot of water drops", "an astronaut rides a horse on mars", "low resolution, blurry, static"]
PROMPTS = ["a cat walks on the beach", "  two   dogs\tplay \n in the snow  ", "a woman smiles &amp; waves &lt;3", "A_Red-Car, DRIVES!!! through the city...",
           "", "unknownword zzz qqq", "the camera pans slowly over a mountain lake at sunset, cinematic lighting, high quality, 4k, slow motion"]


def build_tokenizer():
    from tokenizers import Tokenizer, models, pre_tokenizers, trainers, decoders, processors
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.Unigram())
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.decoder = decoders.Metaspace()
    trainer = trainers.UnigramTrainer(vocab_size=120, special_tokens=["<pad>", "</s>", "<unk>"], unk_token="<unk>")
    tok.train_from_iterator(CORPUS * 4, trainer)
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", eos_token="</s>", unk_token="<unk>")
    os.makedirs(TOK_DIR, exist_ok=True)
    fast.save_pretrained(TOK_DIR)


def main():
    build_tokenizer()
    sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda t: t))          # identity on this fixture's inputs
    spec = importlib.util.spec_from_file_location("ref_tokenizers", os.path.join(REF, "models/wan/modules/tokenizers.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {"prompts": PROMPTS, "cases": {}}
    for clean in ("whitespace", "lower", "canonicalize", None):
        t = m.HuggingfaceTokenizer(TOK_DIR, seq_len=16, clean=clean)
        ids, mask = t(PROMPTS, return_mask=True, add_special_tokens=True)
        out["cases"][str(clean)] = {"ids": ids.tolist(), "mask": mask.tolist(), "cleaned": [t._clean(p) if clean else p for p in PROMPTS]}
    t = m.HuggingfaceTokenizer(TOK_DIR, seq_len=None, clean="whitespace")
    out["cases"]["unpadded_single"] = {"ids": t(PROMPTS[0]).tolist()}
    out["vocab_size"] = t.vocab_size
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "tokenizer.json"), "w"), indent=0)
    print("wrote tokenizer.json; vocab", t.vocab_size, "ids[0] =", out["cases"]["whitespace"]["ids"][0])


if __name__ == "__main__":
    main()
