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
import os as _os

import torch as _torch

# hipGraph replays on ROCm 7.2: with the runtime's "graph packet capture" optimisation (pre-built AQL packets) a replayed graph
# of ~1 000 kernels reads clobbered kernel arguments once a few hundred eager launches have run between two replays -- NaNs,
# or HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; reproduced with plain torch.nn.Linear, none of this package's kernels
# (scripts/graph_bisect.py, scripts/graph_debug3.py; gpurun_out/r2g*).  The knob is read when the HIP runtime initialises,
# so it is set here, at import, unless the user has decided otherwise; lm.GraphedTrainStep refuses to capture when it could
# not take effect.  Eager launches are not affected by it.
# Side effect, stated: this sets a PROCESS-WIDE ROCm runtime knob for every hipGraph user in the process (graphs stay
# correct, launches inside a replay are dispatched the ordinary way); export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 before the
# import to keep the runtime's default -- GraphedTrainStep then refuses to capture.
_K = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_was_init = _torch.cuda.is_initialized()
_prev = _os.environ.get(_K)
_os.environ.setdefault(_K, "0")
# safe = the variable reads "0" AND the HIP runtime reads it after it was set: either it was not initialised yet, or the
# user had exported "0" before starting the process (a value placed in os.environ after initialisation is never seen).
GRAPH_SAFE = _os.environ[_K] == "0" and ((not _was_init) or _prev == "0")

from . import _lib  # noqa: F401,E402

__all__ = ["fftconv", "hyena", "mixer", "filter", "projection", "block", "tokenizer", "build"]
__version__ = "0.1.0"
