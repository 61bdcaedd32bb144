#!/usr/bin/env python
"""Mint golden vectors for the DNA character tokenisation from the REAL reference tokenizer
(``src/dataloaders/datasets/hg38_char_tokenizer.py``, a transformers ``PreTrainedTokenizer`` subclass) called exactly as
``HG38Dataset.__getitem__`` calls it (``src/dataloaders/datasets/hg38_dataset.py:187-223``).

Build container only (``/root/reference`` does not exist on the GPU box):   python oracle/make_golden_tokenizer.py
"""
import os
import random
import sys

import torch

REF = os.environ.get("HYENA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "tokenizer_cases.pt")


def main():
    # load the reference FILE itself (the package __init__ of src.dataloaders pulls in torchvision, which is not installed)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_hg38_char_tokenizer", os.path.join(REF, "src", "dataloaders", "datasets", "hg38_char_tokenizer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    class CharacterTokenizer(mod.CharacterTokenizer):                              # the reference class ...
        """... plus the one method transformers 5.x (installed here; the reference pins 4.26) demands of a slow tokenizer
        during __init__; it returns exactly the table the reference builds at hg38_char_tokenizer.py:57-66."""

        def get_vocab(self):
            specials = ["[CLS]", "[SEP]", "[BOS]", "[MASK]", "[PAD]", "[RESERVED]", "[UNK]"]
            return {**{s: i for i, s in enumerate(specials)}, **{ch: i + 7 for i, ch in enumerate(self.characters)}}
    rng = random.Random(0)
    cases = []
    for max_length, add_eos, replace_n, n_chars, alphabet in [
        (16, True, False, 10, "ACGT"), (16, False, False, 16, "ACGTN"), (16, True, True, 16, "ACGTN"),
        (16, True, False, 40, "ACGT"), (16, False, False, 40, "ACGTN"), (32, True, True, 0, "ACGT"),
        (32, True, False, 31, "ACGTNacgtnXR-"), (32, False, True, 32, "ACGTNacgtn*"), (1026, True, False, 1024, "ACGTN"),
        (1026, True, True, 1026, "ACGTN"), (1026, False, False, 700, "ACGTNn"), (8, True, False, 7, "ACGT"),
        (8, True, False, 8, "ACGT"), (8, False, False, 9, "NNNN"),
    ]:
        # as src/dataloaders/genomics.py:100-104 builds it, minus its inert `add_special_tokens=False` init kwarg, which the
        # transformers 5.x installed here rejects (the reference pins 4.26; the kwarg is only stored there)
        tok = CharacterTokenizer(characters=["A", "C", "G", "T", "N"], model_max_length=max_length + 2, padding_side="left")
        seq = "".join(rng.choice(alphabet) for _ in range(n_chars))
        ids = tok(seq, add_special_tokens=True if add_eos else False, padding="max_length", max_length=max_length,
                  truncation=True)["input_ids"]
        ids = torch.LongTensor(ids)
        if replace_n:
            ids = torch.where(ids == tok._vocab_str_to_int["N"], tok.pad_token_id, ids)
        cases.append({"seq": seq, "max_length": max_length, "add_eos": add_eos, "replace_N_token": replace_n,
                      "data": ids[:-1].clone(), "target": ids[1:].clone()})
    torch.save({"cases": cases, "vocab": dict(tok._vocab_str_to_int), "pad_token_id": tok.pad_token_id,
                "sep_token_id": tok.sep_token_id, "torch": str(torch.__version__)}, OUT)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
