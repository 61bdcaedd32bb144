#!/usr/bin/env python
"""Mint golden samples for the dataset side from the REAL reference classes ``FastaInterval`` / ``HG38Dataset``
(``src/dataloaders/datasets/hg38_dataset.py``) with the reference ``CharacterTokenizer``, on a small synthetic genome.

``pyfaidx`` and ``polars`` are not installed in this image; the reference only uses ``pyfaidx.Fasta`` as "records with a
length that can be sliced into strings" and does not use polars at all, so both are replaced by inert stand-ins here (a
dict-of-strings FASTA parser) -- the interval arithmetic, augmentations, tokenisation and data/target split that are being
pinned all run in the reference's own code.  Build container only:   python oracle/make_golden_dataset.py
"""
import importlib.util
import os
import random
import sys
import types

import torch

REF = os.environ.get("HYENA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "dataset_cases.pt")


class _Record:
    def __init__(self, s):
        self.s = s

    def __len__(self):
        return len(self.s)

    def __getitem__(self, sl):
        return self.s[sl]


class _Fasta(dict):
    def __init__(self, path):
        super().__init__()
        name, parts = None, []
        for line in open(path):
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    self[name] = _Record("".join(parts))
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line)
        if name is not None:
            self[name] = _Record("".join(parts))


def synthetic_genome(rng):
    """three chromosomes, different line widths, lower case and N runs; returned as (fasta text, bed text)"""
    alpha = "ACGT"
    chroms = {"chr1": (5000, 60), "chr2": (3217, 50), "chrX": (901, 70)}
    fa, seqs = [], {}
    for name, (n, width) in chroms.items():
        s = [rng.choice(alpha) for _ in range(n)]
        for _ in range(6):                                  # N runs and soft-masked stretches
            a = rng.randrange(0, n - 40)
            for i in range(a, a + rng.randrange(3, 40)):
                s[i] = "N" if rng.random() < 0.5 else s[i].lower()
        s = "".join(s)
        seqs[name] = s
        fa.append(f">{name} synthetic\n" + "\n".join(s[i:i + width] for i in range(0, n, width)) + "\n")
    bed = []
    for name, (n, _) in chroms.items():
        for j in range(8):
            a = rng.randrange(0, n - 10)
            ln = rng.choice([64, 100, 257, 512, 1000])
            bed.append(f"{name}\t{a}\t{min(n, a + ln)}\t{'train' if j % 4 else 'valid'}")
    bed.append("chr1\t0\t30\ttrain")                        # needs left padding when extended
    bed.append("chr1\t4990\t5000\ttrain")                   # needs right padding
    return "".join(fa), "\n".join(bed) + "\n", seqs


def main():
    sys.modules["pyfaidx"] = types.SimpleNamespace(Fasta=_Fasta)
    sys.modules["polars"] = types.ModuleType("polars")

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ds_mod = load("ref_hg38_dataset", "src/dataloaders/datasets/hg38_dataset.py")
    tok_mod = load("ref_hg38_char_tokenizer", "src/dataloaders/datasets/hg38_char_tokenizer.py")

    class CharacterTokenizer(tok_mod.CharacterTokenizer):      # see oracle/make_golden_tokenizer.py: transformers 5.x needs get_vocab
        def get_vocab(self):
            specials = ["[CLS]", "[SEP]", "[BOS]", "[MASK]", "[PAD]", "[RESERVED]", "[UNK]"]
            return {**{s: i for i, s in enumerate(specials)}, **{ch: i + 7 for i, ch in enumerate(self.characters)}}

    rng = random.Random(7)
    fasta_text, bed_text, _ = synthetic_genome(rng)
    work = os.path.join(HERE, "_ref")
    os.makedirs(work, exist_ok=True)
    fa, bed = os.path.join(work, "golden_genome.fa"), os.path.join(work, "golden_intervals.bed")
    open(fa, "w").write(fasta_text)
    open(bed, "w").write(bed_text)
    cases = []
    for cfg in [dict(max_length=128, add_eos=True, shift_augs=None, rc_aug=False, replace_N_token=False, pad_interval=False),
                dict(max_length=128, add_eos=False, shift_augs=None, rc_aug=False, replace_N_token=True, pad_interval=True),
                dict(max_length=300, add_eos=True, shift_augs=[-5, 7], rc_aug=True, replace_N_token=False, pad_interval=False),
                dict(max_length=1026, add_eos=True, shift_augs=[-40, 40], rc_aug=True, replace_N_token=True, pad_interval=True),
                dict(max_length=64, add_eos=True, shift_augs=None, rc_aug=True, replace_N_token=False, pad_interval=False)]:
        tok = CharacterTokenizer(characters=["A", "C", "G", "T", "N"], model_max_length=cfg["max_length"] + 2, padding_side="left")
        for split in ("train", "valid"):
            ds = ds_mod.HG38Dataset(split, bed, fa, cfg["max_length"], tokenizer=tok, tokenizer_name="char", add_eos=cfg["add_eos"],
                                    shift_augs=cfg["shift_augs"], rc_aug=cfg["rc_aug"], replace_N_token=cfg["replace_N_token"],
                                    pad_interval=cfg["pad_interval"])
            random.seed(1000 + cfg["max_length"])
            samples = [ds[i] for i in range(len(ds))]
            cases.append(dict(cfg=cfg, split=split, seed=1000 + cfg["max_length"], n=len(ds),
                              data=[d for d, _ in samples], target=[t for _, t in samples]))
    torch.save({"fasta": fasta_text, "bed": bed_text, "cases": cases}, OUT)
    print("wrote", OUT, sum(c["n"] for c in cases), "samples")


if __name__ == "__main__":
    main()
