"""The input side of the hot path, second half: sampling intervals of the genome and turning them into
``(data, target)`` token tensors -- ``FastaInterval`` and ``HG38Dataset`` of
``src/dataloaders/datasets/hg38_dataset.py:41-225``, same constructor keywords, same ``__call__`` / ``__getitem__``
contracts, same use of Python's ``random`` for the two augmentations (so a seeded run draws the same shifts / strand
flips as the reference).

What is different underneath:
  * no ``pyfaidx``: ``FastaIndex`` is a small indexed FASTA reader (samtools ``.fai`` layout: name, length, byte offset,
    bases per line, bytes per line -- read from ``<fasta>.fai`` if present, else built by one scan of the file) over a
    memory map; an interval is sliced out with numpy, line breaks removed by a reshape, no Python loop per base;
  * the reverse complement is a 256-entry byte table (the reference appends to a Python string per base,
    hg38_dataset.py:30-38), characters outside ``ACGTacgt`` map to themselves as there;
  * tokenisation is ``hyena_dna_amd.tokenizer.DNACharTokenizerLUT`` (one table lookup) instead of the per-character
    slow-tokenizer loop -- at 10^6 nucleotides per sample the reference's ``__getitem__`` takes seconds, this one
    milliseconds.  Only the ``char`` tokenizer of the hg38 experiments is provided (``bpe`` raises).
Host-side by design (DataLoader workers), like the reference.  ``tests/test_dataset.py`` checks every sample against
vectors minted from the reference classes themselves (``oracle/make_golden_dataset.py``).
"""
import mmap
import os
from pathlib import Path
from random import random, randrange

import numpy as np
import torch

from .tokenizer import DNACharTokenizerLUT

__all__ = ["FastaIndex", "FastaInterval", "HG38Dataset", "reverse_complement"]

_RC = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _RC[_a] = _b


def reverse_complement(seq):
    """bytes -> bytes; ``string_reverse_complement`` (hg38_dataset.py:30-38) as one table lookup"""
    return _RC[np.frombuffer(seq, dtype=np.uint8)[::-1]].tobytes()


def exists(val):
    return val is not None


def coin_flip():
    return random() > 0.5


class FastaIndex:
    """Random access to the records of a FASTA file (what ``pyfaidx.Fasta`` gives the reference: ``keys()``, ``len(f[name])``,
    ``str(f[name][start:end])``)."""

    def __init__(self, fasta_file):
        self.path = str(fasta_file)
        self._open()
        # pyfaidx rebuilds a .fai that is older than its FASTA; a stale or foreign index would silently shift every
        # training sequence.  Use the index only when it is not older than the file AND every record it lists lies inside
        # the file; otherwise index the file itself (one numpy pass).
        fai = self.path + ".fai"
        self.records = None
        if os.path.exists(fai) and os.path.getmtime(fai) >= os.path.getmtime(self.path):
            rec = self._read_fai(fai)
            if rec and all(self._inside(r) for r in rec.values()):
                self.records = rec
        if self.records is None:
            self.records = self._scan()

    def _open(self):
        self._fh = open(self.path, "rb")
        size = os.fstat(self._fh.fileno()).st_size
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ) if size else b""
        self._buf = np.frombuffer(self._mm, dtype=np.uint8) if size else np.zeros(0, np.uint8)

    def _inside(self, rec):
        length, off, lb, lw = rec
        if length < 0 or off < 0 or lb <= 0 or lw < lb:
            return False
        last = off + ((length - 1) // lb) * lw + (length - 1) % lb + 1 if length else off
        return last <= self._buf.shape[0]

    # DataLoader workers started with spawn / forkserver pickle the dataset: the open file, the mmap and the numpy view
    # are per-process state, re-created from the path
    def __getstate__(self):
        return {"path": self.path, "records": self.records}

    def __setstate__(self, state):
        self.path, self.records = state["path"], state["records"]
        self._open()

    @staticmethod
    def _read_fai(path):
        rec = {}
        with open(path) as f:
            for line in f:
                p = line.rstrip("\n").split("\t")
                if len(p) >= 5:
                    rec[p[0]] = tuple(int(x) for x in p[1:5])          # length, offset, bases per line, bytes per line
        return rec

    def _scan(self):
        """one pass over the file: positions of '>' at line starts and of the line feeds, all with numpy"""
        buf = self._buf
        nl = np.flatnonzero(buf == 10)
        starts = np.concatenate(([0], nl + 1))                            # byte offset of every line
        starts = starts[starts < buf.shape[0]]
        heads = starts[buf[starts] == ord(">")]
        rec = {}
        for i, h in enumerate(heads):
            h_end = nl[np.searchsorted(nl, h)] if nl.size and np.searchsorted(nl, h) < nl.size else buf.shape[0]
            name = bytes(buf[h + 1:h_end]).decode("ascii", "replace").split()[0] if h_end > h + 1 else ""
            off = int(h_end) + 1
            end = int(heads[i + 1]) if i + 1 < len(heads) else int(buf.shape[0])
            body_nl = nl[(nl >= off) & (nl < end)]
            first_nl = int(body_nl[0]) if body_nl.size else end
            line_bytes = first_nl - off + 1 if body_nl.size else max(end - off, 1)
            cr = 1 if first_nl > off and body_nl.size and buf[first_nl - 1] == 13 else 0
            line_bases = line_bytes - 1 - cr if body_nl.size else end - off
            # bases = body bytes minus line terminators (a last line with or without one, LF or CRLF)
            length = (end - off) - int(body_nl.size) * (1 + cr)
            rec[name] = (int(length), off, int(max(line_bases, 1)), int(max(line_bytes, 1)))
        return rec

    def keys(self):
        return self.records.keys()

    def length(self, name):
        return self.records[name][0]

    def fetch(self, name, start, end):
        """bases [start, end) of record `name` as bytes (clamped to the record, like slicing a pyfaidx record)"""
        length, off, lb, lw = self.records[name]
        start, end = max(0, min(start, length)), max(0, min(end, length))
        if end <= start:
            return b""
        b0 = off + (start // lb) * lw + start % lb
        b1 = off + ((end - 1) // lb) * lw + (end - 1) % lb + 1
        raw = self._buf[b0:b1]
        if lw == lb or (start // lb) == ((end - 1) // lb):                 # no line break inside the interval
            return raw.tobytes()
        # drop the line terminators: positions (relative to the line grid) >= lb within each lw-byte line
        pos = (np.arange(b0 - off, b1 - off) % lw) < lb
        return raw[pos].tobytes()


class FastaInterval:
    def __init__(self, *, fasta_file, return_seq_indices=False, shift_augs=None, rc_aug=False, pad_interval=False):
        fasta_file = Path(fasta_file)
        assert fasta_file.exists(), "path to fasta file must exist"
        self.seqs = FastaIndex(fasta_file)
        self.return_seq_indices = return_seq_indices
        self.shift_augs = shift_augs
        self.rc_aug = rc_aug
        self.pad_interval = pad_interval
        self.chr_lens = {name: self.seqs.length(name) for name in self.seqs.keys()}

    def fetch_bytes(self, chr_name, start, end, max_length):
        """hg38_dataset.py:70-121 on bytes: shift augmentation, symmetric extension to max_length, clamping, truncation,
        strand flip, optional '.' padding.  Draws from ``random`` in the reference's order (randrange, then random)."""
        interval_length = end - start
        chromosome_length = self.chr_lens[chr_name]
        if exists(self.shift_augs):
            min_shift, max_shift = self.shift_augs
            max_shift += 1
            min_shift = max(start + min_shift, 0) - start
            max_shift = min(end + max_shift, chromosome_length) - end
            rand_shift = randrange(min_shift, max_shift)
            start += rand_shift
            end += rand_shift
        left_padding = right_padding = 0
        if interval_length < max_length:
            extra_seq = max_length - interval_length
            extra_left_seq = extra_seq // 2
            extra_right_seq = extra_seq - extra_left_seq
            start -= extra_left_seq
            end += extra_right_seq
        if start < 0:
            left_padding = -start
            start = 0
        if end > chromosome_length:
            right_padding = end - chromosome_length
            end = chromosome_length
        if interval_length > max_length:
            end = start + max_length
        seq = self.seqs.fetch(chr_name, start, end)
        if self.rc_aug and coin_flip():
            seq = reverse_complement(seq)
        if self.pad_interval:
            seq = (b"." * left_padding) + seq + (b"." * right_padding)
        return seq

    def __call__(self, chr_name, start, end, max_length, return_augs=False):
        return self.fetch_bytes(chr_name, start, end, max_length).decode("ascii", "replace")


class HG38Dataset(torch.utils.data.Dataset):
    """Loop through a bed file, retrieve (chr, start, end), query the fasta file, tokenise (hg38_dataset.py:123-225)."""

    def __init__(self, split, bed_file, fasta_file, max_length, pad_max_length=None, tokenizer=None, tokenizer_name=None,
                 add_eos=False, return_seq_indices=False, shift_augs=None, rc_aug=False, return_augs=False, replace_N_token=False,
                 pad_interval=False):
        self.max_length = max_length
        self.pad_max_length = pad_max_length if pad_max_length is not None else max_length
        self.tokenizer_name = tokenizer_name
        if tokenizer_name not in (None, "char"):
            raise NotImplementedError(f"tokenizer_name={tokenizer_name!r}: the hg38 experiments use 'char' (hg38_hyena.yaml)")
        # the reference passes its CharacterTokenizer here; any object with its alphabet is accepted, only the alphabet is used
        chars = getattr(tokenizer, "characters", None) or ("A", "C", "G", "T", "N")
        side = getattr(tokenizer, "padding_side", "left")
        self.tokenizer = tokenizer
        self.lut = DNACharTokenizerLUT(characters=tuple(chars), padding_side=side)
        self.return_augs = return_augs
        self.add_eos = add_eos
        self.replace_N_token = replace_N_token
        self.pad_interval = pad_interval
        bed_path = Path(bed_file)
        assert bed_path.exists(), "path to .bed file must exist"
        rows = []
        with open(bed_path) as f:                                   # chr_name, start, end, split (tab separated, no header)
            for line in f:
                p = line.rstrip("\n").split("\t")
                if len(p) >= 4 and p[3] == split:
                    rows.append((p[0], int(p[1]), int(p[2])))
        self.rows = rows
        self.fasta = FastaInterval(fasta_file=fasta_file, return_seq_indices=return_seq_indices, shift_augs=shift_augs,
                                   rc_aug=rc_aug, pad_interval=pad_interval)

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        chr_name, start, end = self.rows[idx]
        seq = self.fasta.fetch_bytes(chr_name, start, end, max_length=self.max_length)
        return self.lut.sample(seq, self.max_length, add_eos=self.add_eos, replace_N_token=self.replace_N_token)
