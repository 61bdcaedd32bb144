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


def test_stale_or_foreign_fai_is_not_trusted_and_index_pickles(genome, tmp_path):
    """pyfaidx rebuilds an index older than its FASTA; spawn-started DataLoader workers pickle the dataset."""
    import pickle
    import shutil
    import time
    g, fa, _ = genome
    fa2 = str(tmp_path / "g.fa")
    shutil.copy(fa, fa2)
    good = FastaIndex(fa2)
    name = next(iter(good.keys()))
    length, off, lb, lw = good.records[name]
    want = good.fetch(name, 3, 200)
    # (a) an index that points outside the file (foreign)
    with open(fa2 + ".fai", "w") as f:
        f.write(f"{name}\t{length}\t{off + 10 ** 9}\t{lb}\t{lw}\n")
    assert FastaIndex(fa2).fetch(name, 3, 200) == want
    # (b) an index older than the FASTA with plausible but wrong offsets (stale)
    with open(fa2 + ".fai", "w") as f:
        f.write(f"{name}\t{length}\t{off + 1}\t{lb}\t{lw}\n")
    past = time.time() - 1000
    os.utime(fa2 + ".fai", (past, past))
    assert FastaIndex(fa2).fetch(name, 3, 200) == want
    # (c) a fresh, correct index is used as is
    with open(fa2 + ".fai", "w") as f:
        for n, r in good.records.items():
            f.write("\t".join([n] + [str(x) for x in r]) + "\n")
    idx = FastaIndex(fa2)
    assert idx.records == good.records and idx.fetch(name, 3, 200) == want
    # pickling re-opens the file in the receiving process
    clone = pickle.loads(pickle.dumps(idx))
    assert clone.fetch(name, 3, 200) == want and clone.records == idx.records
    ds = FastaInterval(fasta_file=fa2, return_seq_indices=False, shift_augs=None, rc_aug=False)
    assert pickle.loads(pickle.dumps(ds)).seqs.fetch(name, 3, 200) == want
