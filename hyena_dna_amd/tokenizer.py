"""Vectorised DNA character tokenisation: the input side of the hot path.

Mirrors what ``HG38Dataset.__getitem__`` (``src/dataloaders/datasets/hg38_dataset.py:187-223``) gets out of the reference's
``CharacterTokenizer`` (``src/dataloaders/datasets/hg38_char_tokenizer.py``: specials ``[CLS]=0 [SEP]=1 [BOS]=2 [MASK]=3
[PAD]=4 [RESERVED]=5 [UNK]=6``, then one id per character starting at 7 -- ``A C G T N`` -> 7..11 for hg38,
``src/dataloaders/genomics.py:100-104``) when called with ``padding="max_length", truncation=True`` and
``add_special_tokens=add_eos``: truncate to ``max_length`` (leaving room for the ``[SEP]`` that serves as EOS), map
characters, append ``[SEP]``, pad on the LEFT with ``[PAD]``; optionally replace ``N`` by ``[PAD]`` so the loss ignores it;
``data = ids[:-1]``, ``target = ids[1:]``.

The reference does this through the generic slow-tokenizer machinery of ``transformers`` -- a Python-level loop over
every character (``_tokenize`` = ``list(text)``, one dict lookup per token), seconds per 10^6-nucleotide sample.  Here it is
one 256-entry byte table lookup (numpy), ~1 ms per 10^6 nucleotides, same ids bit for bit (``tests/test_tokenizer.py``
against vectors minted from the reference class, ``oracle/make_golden_tokenizer.py``).  Host-side by design: samples are
produced by DataLoader workers on the CPU, as in the reference.
"""
import numpy as np
import torch

__all__ = ["DNACharTokenizerLUT"]

SPECIALS = ("[CLS]", "[SEP]", "[BOS]", "[MASK]", "[PAD]", "[RESERVED]", "[UNK]")


class DNACharTokenizerLUT:
    def __init__(self, characters=("A", "C", "G", "T", "N"), padding_side="left"):
        if padding_side not in ("left", "right"):
            raise ValueError(f"padding_side must be 'left' or 'right', got {padding_side!r}")
        self.characters = tuple(characters)
        self.padding_side = padding_side
        self.vocab = {**{s: i for i, s in enumerate(SPECIALS)}, **{ch: i + 7 for i, ch in enumerate(self.characters)}}
        self.cls_token_id, self.sep_token_id, self.bos_token_id, self.mask_token_id = 0, 1, 2, 3
        self.pad_token_id, self.unk_token_id = 4, 6
        self.eos_token_id = self.sep_token_id
        lut = np.full(256, self.unk_token_id, dtype=np.int64)
        for ch in self.characters:
            if len(ch) != 1 or ord(ch) > 127:
                raise ValueError(f"characters must be single ASCII characters, got {ch!r}")
            lut[ord(ch)] = self.vocab[ch]
        self._lut = lut

    @property
    def vocab_size(self):
        return len(self.vocab)

    def encode(self, seq, max_length, add_eos=True):
        """ids (max_length,) int64 numpy: what ``tokenizer(seq, add_special_tokens=add_eos, padding="max_length",
        max_length=max_length, truncation=True)["input_ids"]`` returns."""
        raw = seq if isinstance(seq, (bytes, bytearray, memoryview)) else seq.encode("ascii", "replace")   # non-ASCII -> '?' -> [UNK]
        keep = max(0, max_length - (1 if add_eos else 0))
        body = self._lut[np.frombuffer(raw, dtype=np.uint8, count=min(len(raw), keep))]
        out = np.full(max_length, self.pad_token_id, dtype=np.int64)
        n = body.shape[0] + (1 if add_eos and max_length > 0 else 0)
        start = max_length - n if self.padding_side == "left" else 0
        out[start:start + body.shape[0]] = body
        if add_eos and max_length > 0:
            out[start + body.shape[0]] = self.sep_token_id
        return out

    def sample(self, seq, max_length, add_eos=True, replace_N_token=False):
        """(data, target) LongTensors of length max_length - 1, as HG38Dataset.__getitem__ returns them."""
        ids = torch.from_numpy(self.encode(seq, max_length, add_eos))
        if replace_N_token and "N" in self.vocab:
            ids = torch.where(ids == self.vocab["N"], self.pad_token_id, ids)
        return ids[:-1].clone(), ids[1:].clone()

    def decode(self, ids):
        inv = {v: k for k, v in self.vocab.items()}
        return "".join(inv[int(i)] for i in ids)
