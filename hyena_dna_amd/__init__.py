"""hyena_dna_amd -- MI355X-native Hyena long convolution (HIP/gfx950) behind the reference's own surface.

Scope: the hot path ``HyenaOperator -> HyenaFilter -> fftconv`` of
HazyResearch/hyena-dna, nothing else.

* ``hyena_dna_amd.fftconv``  mirrors ``src/ops/fftconv.py``      (``fftconv_func``, ``FFTConvFunc``)
* ``hyena_dna_amd.hyena``    mirrors ``src/models/sequence/hyena.py`` (``HyenaOperator``, ``HyenaFilter``)
* ``hyena_dna_amd.mixer``    the fused core of ``HyenaOperator.forward`` between the projections (hyena.py:392-439)
* ``hyena_dna_amd.filter``   the fused implicit filter (``HyenaFilter.filter``, hyena.py:229-238)
* ``hyena_dna_amd.projection``  ``in_proj`` / ``out_proj`` with a slice-batched weight gradient
* ``hyena_dna_amd.block``    ``dropout_add_layer_norm`` (the block's residual add + LayerNorm, fused)
* ``hyena_dna_amd.tokenizer``  vectorised DNA character tokenisation (the input side)
* ``hyena_dna_amd.csrc``     the HIP kernels and the C ABI (``include/hyena_fftconv.h``, ``hyena_mixer.h``, ``hyena_filter.h``)

There is no CPU or PyTorch fallback for the convolution: without the compiled gfx950 library the ops raise.
"""
from . import _lib  # noqa: F401

__all__ = ["fftconv", "hyena", "mixer", "filter", "projection", "block", "tokenizer", "build"]
__version__ = "0.1.0"
