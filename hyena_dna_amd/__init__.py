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
# (scripts/graph_bisect.py, scripts/graph_debug3.py; gpurun_out/r2g*).  Whole-step replays (lm.GraphedTrainStep) therefore need
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, and the knob is read when the HIP runtime initialises.
#
# Round 4: importing this package no longer touches the environment (the knob is process-wide, and most users of
# the package -- the overlay seams, eager training -- never capture a graph).  Whoever wants graphed steps says so, BEFORE the first
# torch.cuda call: `hyena_dna_amd.prepare_graph_runtime()` (or export the variable; bench.py, scripts/train_hg38.py and the tests set it
# at their very top).  GraphedTrainStep refuses to capture when the knob could not have taken effect.  Eager launches never need it.
_K = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
# safe = the variable reads "0" AND the HIP runtime reads it after it was set: found "0" at import (exported before the process started,
# or set by the caller ahead of its imports), or set by prepare_graph_runtime() while the runtime was still uninitialised
# (a process that starts the HIP runtime, THEN puts "0" into os.environ, THEN imports this package must not pass -- the runtime
# never saw the knob.  A variable exported before the process started is indistinguishable from that here only if the runtime is already
# up at import, and then the conservative answer is "not safe": export HYENA_GRAPH_SAFE_OVERRIDE=1 to vouch for it.)
GRAPH_SAFE = _os.environ.get(_K) == "0" and (not _torch.cuda.is_initialized() or _os.environ.get("HYENA_GRAPH_SAFE_OVERRIDE") == "1")


def prepare_graph_runtime():
    """Opt in to hipGraph replays of whole training steps: sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for this process if the HIP runtime has
    not been initialised yet and the user has not chosen a value.  Returns whether graphed steps are safe to use."""
    global GRAPH_SAFE
    if _os.environ.get(_K) == "0":
        return GRAPH_SAFE
    if _K in _os.environ or _torch.cuda.is_initialized():      # the user decided otherwise / too late to take effect
        return False
    _os.environ[_K] = "0"
    GRAPH_SAFE = True
    return True


from . import _lib  # noqa: F401,E402

__all__ = ["fftconv", "hyena", "mixer", "filter", "projection", "block", "tokenizer", "build", "prepare_graph_runtime", "GRAPH_SAFE"]
__version__ = "0.1.0"
