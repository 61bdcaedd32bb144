"""Import surface of ``flash_attn`` that the UNMODIFIED reference backbone needs (src/models/sequence/long_conv_lm.py:18-33,
dna_embedding.py:5-9), served by this repository's MI355X-side model glue (hyena_dna_amd.lm / hyena_dna_amd.block).

Put ``<repo>/overlay`` ahead of the reference on ``sys.path`` (INTEGRATION.md sections 1 and 4); the CUDA-only flash_attn
wheel is then not needed on ROCm.  ``flash_attn.ops.fused_dense`` and ``flash_attn.losses`` are deliberately absent: the
reference guards those imports (long_conv_lm.py:25-28, hyena.py:18-21, tasks/torchmetrics.py:13-16) and falls back to
its own PyTorch code."""
__version__ = "0.0+hyena_dna_amd"
