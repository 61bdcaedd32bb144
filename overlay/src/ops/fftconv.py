"""Path overlay for the UNMODIFIED reference tree.

``src`` and ``src.ops`` are PEP-420 namespace packages in the reference, so putting this directory ahead of the
reference on ``sys.path`` makes ``src.ops.fftconv`` resolve here while everything else still loads from the
reference:

    cd <rundir> && PYTHONPATH=<repo>/overlay:<repo>:/path/to/hyena-dna python -m train \
        experiment=hg38/hg38_hyena model.layer.fused_fft_conv=true ...

The reference's ``src/models/sequence/hyena.py:12-16`` import then succeeds (``fftconv_func`` is no longer None) and
``HyenaFilter.forward`` (hyena.py:250-259) routes every long convolution through the MI355X HIP kernels.
"""
from hyena_dna_amd.fftconv import FFTConvFunc, fftconv_func, fftconv_heads_ref, fftconv_ref  # noqa: F401
