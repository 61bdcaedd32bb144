"""hyena_dna_amd.dataset (FastaIndex / FastaInterval / HG38Dataset) against samples minted from the reference's own classes
(oracle/make_golden_dataset.py: src/dataloaders/datasets/hg38_dataset.py:41-225 + the reference CharacterTokenizer) on a
synthetic genome: every (data, target) pair identical, augmentations drawn from `random` in the reference's order."""
import os
import random

import pytest
import torch

from hyena_dna_amd.dataset import FastaIndex, FastaInterval, HG38Dataset, reverse_complement

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_cases.pt")


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    g = torch.load(GOLDEN, weights_only=False)
    d = tmp_path_factory.mktemp("genome")
    fa, bed = d / "genome.fa", d / "intervals.bed"
    fa.write_text(g["fasta"])
    bed.write_text(g["bed"])
    return g, str(fa), str(bed)


def test_samples_match_the_reference_dataset(genome):
    g, fa, bed = genome
    total = 0
    for c in g["cases"]:
        cfg = c["cfg"]
        ds = HG38Dataset(c["split"], bed, fa, cfg["max_length"], tokenizer=None, tokenizer_name="char", add_eos=cfg["add_eos"],
                         shift_augs=cfg["shift_augs"], rc_aug=cfg["rc_aug"], replace_N_token=cfg["replace_N_token"],
                         pad_interval=cfg["pad_interval"])
        assert len(ds) == c["n"]
        random.seed(c["seed"])
        for i in range(len(ds)):
            data, target = ds[i]
            assert data.dtype == torch.int64 and torch.equal(data, c["data"][i]) and torch.equal(target, c["target"][i]), (cfg, c["split"], i)
            total += 1
    assert total == 130


def test_fasta_index_matches_plain_parsing(genome, tmp_path):
    g, fa, _ = genome
    seqs, name = {}, None
    for line in g["fasta"].splitlines():
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        else:
            seqs[name] += line
    idx = FastaIndex(fa)
    assert set(idx.keys()) == set(seqs)
    rng = random.Random(3)
    for name, s in seqs.items():
        assert idx.length(name) == len(s)
        for _ in range(200):
            a = rng.randrange(0, len(s))
            b = rng.randrange(a, min(len(s), a + 400) + 1)
            assert idx.fetch(name, a, b) == s[a:b].encode()
        assert idx.fetch(name, len(s) - 5, len(s) + 50) == s[-5:].encode() and idx.fetch(name, 10, 10) == b""
    # a samtools-style .fai next to the file is used instead of scanning (same answers)
    fa2 = tmp_path / "g2.fa"
    fa2.write_text(g["fasta"])
    with open(str(fa2) + ".fai", "w") as f:
        for name, (length, off, lb, lw) in idx.records.items():
            f.write(f"{name}\t{length}\t{off}\t{lb}\t{lw}\n")
    idx2 = FastaIndex(str(fa2))
    assert idx2.records == idx.records and idx2.fetch("chr2", 7, 333) == seqs["chr2"][7:333].encode()
    # CRLF line ends and a last line without terminator
    fa3 = tmp_path / "g3.fa"
    fa3.write_bytes(b">a x\r\nACGTAC\r\nGTTT\r\n>b\r\nNNAC")
    idx3 = FastaIndex(str(fa3))
    assert idx3.length("a") == 10 and idx3.fetch("a", 4, 9) == b"ACGTT" and idx3.length("b") == 4 and idx3.fetch("b", 1, 4) == b"NAC"


def test_reverse_complement_and_interval_call(genome):
    g, fa, _ = genome
    assert reverse_complement(b"ACGTNacgt-x") == b"x-acgtNACGT"
    fi = FastaInterval(fasta_file=fa, pad_interval=True)
    s = fi("chr1", 0, 30, max_length=64)
    assert isinstance(s, str) and len(s) == 64 and s.startswith("." * 17) and not s.endswith(".")
    with pytest.raises(NotImplementedError):
        HG38Dataset("train", genome[2], fa, 64, tokenizer_name="bpe")
