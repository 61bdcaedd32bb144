"""ctypes binding of the C ABI in include/hyena_fftconv.h (libhyena_fftconv.so, built by hyena_dna_amd.build).

The library is loaded lazily on first use.  There is deliberately no fallback: if the shared object is missing
or a call returns a non-zero status the caller gets an exception, never a silently different code path.
"""
import ctypes
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libhyena_fftconv.so")
ABI_VERSION = 3

HYENA_F32, HYENA_BF16, HYENA_F16 = 0, 1, 2
MAX_L = 1048576
PLAN_NONE, PLAN_ONCHIP, PLAN_TWO_LEVEL = 0, 1, 2

_DTYPES = {torch.float32: HYENA_F32, torch.bfloat16: HYENA_BF16, torch.float16: HYENA_F16}

_lock = threading.Lock()
_lib = None
_tables = {}      # (device index, M) -> uint8 tensor holding the twiddle tables
_workspace = {}   # (device index, stream) -> uint8 tensor


class HyenaLibraryError(RuntimeError):
    pass


class FilterParams(ctypes.Structure):
    """hyena_filter_params of include/hyena_filter.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("z", "t", "w0", "b0", "w1", "b1", "w2", "b2", "w3", "freq", "deltas")] + \
               [("shift", ctypes.c_float), ("modulate", ctypes.c_int), ("z_stride", ctypes.c_int),
                ("L", ctypes.c_int), ("E", ctypes.c_int), ("D", ctypes.c_int)]


class FilterGrads(ctypes.Structure):
    """hyena_filter_grads of include/hyena_filter.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("dw0", "db0", "dw1", "db1", "dw2", "db2", "dw3", "dfreq", "dz")]


class _HipBackend:
    """Where tensors live and how the library is reached.  The product has exactly this one backend (ROCm device
    tensors + libhyena_fftconv.so); tests/ substitute a CPU-emulation double to exercise the host logic without
    a GPU (tests/hipemu/emu_backend.py) -- nothing in this package ever selects another backend."""
    name = "hip"
    path = os.environ.get("HYENA_FFTCONV_LIB", LIB_PATH)      # (development: A/B builds of the same HIP library)

    def require(self, t, name):
        if not t.is_cuda:
            raise HyenaLibraryError(
                f"hyena fftconv: `{name}` lives on {t.device}; the HIP kernels need a ROCm device tensor "
                "(there is no CPU fallback)")

    def guard(self, device):
        return torch.cuda.device(device)

    def stream(self, device):
        return torch.cuda.current_stream(device).cuda_stream

    def capturing(self):
        return torch.cuda.is_current_stream_capturing()

    def free_memory(self, device=None):
        """Bytes a new allocation on `device` can take: what the driver reports free plus what PyTorch's caching
        allocator holds but is not using (it hands those blocks out again without asking the driver)."""
        if torch.cuda.is_current_stream_capturing():      # no driver queries inside a hipGraph capture
            return None
        return (torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device)
                - torch.cuda.memory_allocated(device))


_backend = _HipBackend()


def lib():
    """The loaded shared library; raises HyenaLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        LIB_PATH = _backend.path
        if not os.path.exists(LIB_PATH):
            raise HyenaLibraryError(
                f"{LIB_PATH} not found: build it with `python -m hyena_dna_amd.build` (hipcc, gfx950). "
                "hyena_dna_amd has no CPU/PyTorch fallback for the long convolution.")
        L = ctypes.CDLL(LIB_PATH)
        c_int, c_size_t, c_void_p, c_char_p = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p
        L.hyena_fftconv_abi_version.restype = c_int
        L.hyena_fftconv_abi_version.argtypes = []
        L.hyena_fftconv_error_string.restype = c_char_p
        L.hyena_fftconv_error_string.argtypes = [c_int]
        L.hyena_fftconv_fft_size.restype = c_int
        L.hyena_fftconv_fft_size.argtypes = [c_int]
        L.hyena_fftconv_plan.restype = c_int
        L.hyena_fftconv_plan.argtypes = [c_int]
        L.hyena_fftconv_table_bytes.restype = c_size_t
        L.hyena_fftconv_table_bytes.argtypes = [c_int]
        L.hyena_fftconv_init_tables.restype = c_int
        L.hyena_fftconv_init_tables.argtypes = [c_void_p, c_int, c_void_p]
        L.hyena_fftconv_default_chunk.restype = c_int
        L.hyena_fftconv_default_chunk.argtypes = [c_int, c_int, c_int, c_int]
        L.hyena_fftconv_workspace_bytes.restype = c_size_t
        L.hyena_fftconv_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
        L.hyena_fftconv_fwd.restype = c_int
        L.hyena_fftconv_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        L.hyena_fftconv_bwd.restype = c_int
        L.hyena_fftconv_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        L.hyena_fftconv_saved_bytes.restype = c_size_t
        L.hyena_fftconv_saved_bytes.argtypes = [c_int, c_int, c_int]
        L.hyena_fftconv_fwd_save.restype = c_int
        L.hyena_fftconv_fwd_save.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                             c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p]
        L.hyena_fftconv_bwd_saved.restype = c_int
        L.hyena_fftconv_bwd_saved.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                              c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p]
        L.hyena_fftconv_fwd_ld.restype = c_int
        L.hyena_fftconv_fwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p]
        L.hyena_fftconv_bwd_ld.restype = c_int
        L.hyena_fftconv_bwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_int,
                                           c_void_p, c_size_t, c_void_p]
        # fused mixer shell (include/hyena_mixer.h)
        L.hyena_mixer_pre_fwd.restype = c_int
        L.hyena_mixer_pre_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_mixer_post_fwd.restype = c_int
        L.hyena_mixer_post_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p]
        L.hyena_mixer_partial_floats.restype = c_size_t
        L.hyena_mixer_partial_floats.argtypes = [c_int, c_int, c_int]
        L.hyena_mixer_post_bwd.restype = c_int
        L.hyena_mixer_post_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_mixer_pre_bwd.restype = c_int
        L.hyena_mixer_pre_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_int, c_int, c_int, c_void_p]
        # the same shell in channel-major layout (csrc/cm_kernels.h)
        L.hyena_cm_partial_floats.restype = c_size_t
        L.hyena_cm_partial_floats.argtypes = [c_int, c_int, c_int]
        L.hyena_cm_pre_fwd.restype = c_int
        L.hyena_cm_pre_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_cm_post_fwd.restype = c_int
        L.hyena_cm_post_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p]
        L.hyena_cm_post_bwd.restype = c_int
        L.hyena_cm_post_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_cm_pre_bwd.restype = c_int
        L.hyena_cm_pre_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_cm_pre_fwd_ld.restype = c_int
        L.hyena_cm_pre_fwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_long, c_int, c_int,
                                          c_int, c_void_p]
        L.hyena_cm_post_fwd_ld.restype = c_int
        L.hyena_cm_post_fwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_long, c_int,
                                           ctypes.c_long, c_int, c_int, c_int, c_void_p]
        L.hyena_cm_post_bwd_ld.restype = c_int
        L.hyena_cm_post_bwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, ctypes.c_long, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        L.hyena_cm_pre_bwd_ld.restype = c_int
        L.hyena_cm_pre_bwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_int, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        # input projection on the matrix cores + front of the shell (include/hyena_proj.h)
        L.hyena_inproj_pre_fwd_ld.restype = c_int
        L.hyena_inproj_pre_fwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int, c_int, c_int, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        L.hyena_outproj_gate_addnorm_fwd_ld.restype = c_int
        L.hyena_outproj_gate_addnorm_fwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                        c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                        c_int, c_int, c_int, c_int, ctypes.c_long, c_int, ctypes.c_long, c_int, c_int, c_int,
                                                        c_void_p]
        L.hyena_outproj_dgrad_supported.restype = c_int
        L.hyena_outproj_dgrad_supported.argtypes = [c_int, c_int, c_int, c_int]
        L.hyena_outproj_dgrad_partial_floats.restype = c_size_t
        L.hyena_outproj_dgrad_partial_floats.argtypes = [c_int, c_int, c_int]
        L.hyena_outproj_dgrad_gate_bwd_ld.restype = c_int
        L.hyena_outproj_dgrad_gate_bwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                      c_void_p, c_int, c_int, c_int, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        L.hyena_outproj_gate_fwd_ld.restype = c_int
        L.hyena_outproj_gate_fwd_ld.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_int, c_int, c_int, c_int, ctypes.c_long, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        L.hyena_proj_supported.restype = c_int
        L.hyena_proj_supported.argtypes = [c_int, c_int, c_int, c_int]
        L.hyena_inproj_pre_fwd.restype = c_int
        L.hyena_inproj_pre_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_outproj_supported.restype = c_int
        L.hyena_outproj_supported.argtypes = [c_int, c_int, c_int, c_int]
        L.hyena_proj_kernel_generation.restype = c_int
        L.hyena_proj_kernel_generation.argtypes = [c_int, c_int]
        gen = os.environ.get("HYENA_OUTPROJ_KERNEL")              # A/B knob: 1 = round 4's out_proj kernel, 2 (default) = round 6's
        if gen in ("1", "2"):
            L.hyena_proj_kernel_generation(0, int(gen))
        gen = os.environ.get("HYENA_INPROJ_KERNEL")               # A/B knob: 1 (default) = rounds 3 / 4's in_proj kernel, 2 = round 6's (not faster: see the header)
        if gen in ("1", "2"):
            L.hyena_proj_kernel_generation(1, int(gen))
        L.hyena_outproj_gate_fwd.restype = c_int
        L.hyena_outproj_gate_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_int, c_int, c_int, c_int, c_void_p]
        L.hyena_colsum_supported.restype = c_int
        L.hyena_colsum_supported.argtypes = [ctypes.c_long, c_int, c_int]
        L.hyena_colsum_partial_floats.restype = c_size_t
        L.hyena_colsum_partial_floats.argtypes = [ctypes.c_long, c_int]
        L.hyena_colsum.restype = c_int
        L.hyena_colsum.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_void_p]
        L.hyena_mlp_supported.restype = c_int
        L.hyena_mlp_supported.argtypes = [ctypes.c_long, c_int, c_int, c_int]
        L.hyena_mlp_partial_floats.restype = c_size_t
        L.hyena_mlp_partial_floats.argtypes = [ctypes.c_long, c_int]
        L.hyena_mlp_fc1_gelu_fwd.restype = c_int
        L.hyena_mlp_fc1_gelu_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        L.hyena_mlp_dh_dgelu_bwd.restype = c_int
        L.hyena_mlp_dh_dgelu_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_void_p]
        # fused implicit filter (include/hyena_filter.h)
        L.hyena_filter_supported.restype = c_int
        L.hyena_filter_supported.argtypes = [c_int, c_int, c_int, c_int]
        L.hyena_filter_saved_bytes.restype = c_size_t
        L.hyena_filter_saved_bytes.argtypes = [c_int]
        L.hyena_filter_workspace_bytes.restype = c_size_t
        L.hyena_filter_workspace_bytes.argtypes = [c_int, c_int]
        L.hyena_filter_fwd.restype = c_int
        L.hyena_filter_fwd.argtypes = [ctypes.POINTER(FilterParams), c_void_p, c_void_p, c_void_p]
        L.hyena_filter_bwd.restype = c_int
        L.hyena_filter_bwd.argtypes = [ctypes.POINTER(FilterParams), c_void_p, c_void_p, ctypes.POINTER(FilterGrads),
                                       c_void_p, c_size_t, c_void_p]
        L.hyena_filter16_saved_bytes.restype = c_size_t
        L.hyena_filter16_saved_bytes.argtypes = [c_int]
        L.hyena_filter16_fwd.restype = c_int
        L.hyena_filter16_fwd.argtypes = [ctypes.POINTER(FilterParams), c_int, c_void_p, c_void_p, c_void_p]
        L.hyena_filter16_bwd.restype = c_int
        L.hyena_filter16_bwd.argtypes = [ctypes.POINTER(FilterParams), c_int, c_void_p, c_void_p, ctypes.POINTER(FilterGrads),
                                         c_void_p, c_size_t, c_void_p]
        L.hyena_filter_row_pitch.restype = c_int
        L.hyena_filter_row_pitch.argtypes = [c_int]
        L.hyena_filter_fwd_ld.restype = c_int
        L.hyena_filter_fwd_ld.argtypes = [ctypes.POINTER(FilterParams), c_void_p, c_int, c_void_p, c_void_p]
        L.hyena_filter_bwd_ld.restype = c_int
        L.hyena_filter_bwd_ld.argtypes = [ctypes.POINTER(FilterParams), c_void_p, c_int, c_void_p, ctypes.POINTER(FilterGrads),
                                          c_void_p, c_size_t, c_void_p]
        L.hyena_filter16_fwd_ld.restype = c_int
        L.hyena_filter16_fwd_ld.argtypes = [ctypes.POINTER(FilterParams), c_int, c_void_p, c_int, c_void_p, c_void_p]
        L.hyena_filter16_bwd_ld.restype = c_int
        L.hyena_filter16_bwd_ld.argtypes = [ctypes.POINTER(FilterParams), c_int, c_void_p, c_int, c_void_p, ctypes.POINTER(FilterGrads),
                                            c_void_p, c_size_t, c_void_p]
        # fused residual add + LayerNorm (include/hyena_block.h)
        c_long, c_float = ctypes.c_long, ctypes.c_float
        L.hyena_add_norm_supported.restype = c_int
        L.hyena_add_norm_supported.argtypes = [c_int, c_int, c_int]
        L.hyena_add_norm_fwd.restype = c_int
        L.hyena_add_norm_fwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p,
                                         c_void_p, c_void_p, c_long, c_int, c_void_p]
        L.hyena_dropout_add_norm_fwd.restype = c_int
        L.hyena_dropout_add_norm_fwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int,
                                                 c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]
        L.hyena_dropout_add_norm_bwd.restype = c_int
        L.hyena_dropout_add_norm_bwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]
        L.hyena_dropout_add_norm_bwd_colsum.restype = c_int
        L.hyena_dropout_add_norm_bwd_colsum.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                                        c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]
        L.hyena_embed_add_norm_supported.restype = c_int
        L.hyena_embed_add_norm_supported.argtypes = [c_int, c_int, c_int]
        L.hyena_embed_add_norm_fwd.restype = c_int
        L.hyena_embed_add_norm_fwd.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]
        L.hyena_embed_add_norm_partial_floats.restype = c_size_t
        L.hyena_embed_add_norm_partial_floats.argtypes = [c_long, c_int]
        L.hyena_embed_add_norm_bwd.restype = c_int
        L.hyena_embed_add_norm_bwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                               c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]
        L.hyena_add_norm_partial_floats.restype = c_size_t
        L.hyena_add_norm_partial_floats.argtypes = [c_long, c_int]
        L.hyena_add_norm_bwd.restype = c_int
        L.hyena_add_norm_bwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]
        if L.hyena_fftconv_abi_version() != ABI_VERSION:
            raise HyenaLibraryError(f"{LIB_PATH}: ABI version {L.hyena_fftconv_abi_version()} != {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise HyenaLibraryError("hyena_fftconv: " + lib().hyena_fftconv_error_string(status).decode())


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise TypeError(f"hyena fftconv supports float32 / bfloat16 / float16 activations, got {dtype}") from None


def _require_gpu(t, name):
    _backend.require(t, name)


# ---- pitched rows (round 5) ---------------------------------------------------------------------------------------------------------
# The reference's trainer hands the operator L = max_length - 1 positions (hg38_dataset.py:220-223): odd.  In a packed channel-major tensor
# every second row of 16-bit elements then starts 2 bytes off a 4-byte boundary and none but the first is 16-byte aligned, which the
# kernels' 16-byte accesses, the library GEMMs' vector loads and the filter kernels' fast paths all pay for (21 % of a layer at 2^20 - 1,
# profiles/r5a_*).  So every tensor this package allocates BETWEEN the two projections -- xT, vg, y, zT, their gradients, the filter k and
# dk -- is a [..., :L] view of a buffer whose rows are ROW_ALIGN elements apart (L rounded up): rows start 128-byte (16-bit) / 256-byte
# (fp32) aligned whatever L is, and the C ABI's *_ld entry points are told the pitch.  With L a multiple of ROW_ALIGN nothing changes.
ROW_ALIGN = 64


def row_pitch(L):
    return (int(L) + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN


def empty_rows(lead, L, dtype, device, pitch=None):
    """An uninitialised tensor of shape (*lead, L) whose rows are `pitch` (default row_pitch(L)) elements apart: the [..., :L] view of a
    packed (*lead, pitch) buffer."""
    ld = row_pitch(L) if pitch is None else int(pitch)
    buf = torch.empty(tuple(lead) + (ld,), dtype=dtype, device=device)
    return buf if ld == L else buf[..., :L]


def ld_of(t):
    """Row pitch (elements) of a tensor whose last dimension is contiguous and whose leading dimensions are packed over equally pitched
    rows -- what empty_rows hands out, and every contiguous tensor (pitch = L); None for any other layout."""
    if t.dim() < 1:
        return None
    L = t.shape[-1]
    if L > 1 and t.stride(-1) != 1:
        return None
    if t.dim() == 1:
        return L
    # the pitch is the stride of the innermost leading dimension with more than one entry; all others must be packed over it
    ld, expect = None, None
    for i in range(t.dim() - 2, -1, -1):
        if t.shape[i] == 1:
            continue
        if ld is None:
            ld = t.stride(i)
            if ld < L:
                return None
            expect = ld * t.shape[i]
        else:
            if t.stride(i) != expect:
                return None
            expect *= t.shape[i]
    return L if ld is None else ld


def as_rows(t):
    """t itself if its layout is pitched rows (ld_of), else a packed copy.  A pitched view of ODD length must also own one element behind its last row:
    on pitched rows the two-level plan takes a row's last sample with its regular pair load (csrc/fftconv_kernels.h, tail_pair) and drops the element
    behind it by a select -- that element has to be readable (a caller's view such as buf[..., 1:] of a (B, D, L + 1) tensor ends exactly
    at its storage's end; empty_rows' buffers never do)."""
    ld = ld_of(t)
    if ld is None:
        return t.contiguous()
    L = t.shape[-1] if t.dim() else 0
    if t.dim() >= 2 and ld > L and (L & 1) and t.numel() > 0:
        rows = t.numel() // L
        last = t.storage_offset() + (rows - 1) * ld + L              # first element behind the last row
        if last >= t.untyped_storage().nbytes() // t.element_size():
            return t.contiguous()
    return t


def empty_like_rows(t, dtype=None):
    """A new tensor with t's shape and t's row pitch"""
    return empty_rows(t.shape[:-1], t.shape[-1], t.dtype if dtype is None else dtype, t.device, pitch=ld_of(t))


# Channel-major (C, B, L) tensors that a library GEMM touches as well -- xT, dxT, zT, dzT -- come in a second layout when B > 1: the CHANNEL rows
# are pitched over the flattened positions (row (c, b) at c cs + b L with cs = B L rounded up), so that the GEMM still sees ONE (C, B L) matrix
# with a leading dimension (per-sequence pitched rows would need one product per sequence: 32767 x 8 ran 64 % slower that way,
# profiles/r5h_*) and the in_proj kernel's 64-position tiles over the flattened positions store aligned.  For B = 1 the two layouts coincide.
def empty_cm(C, B, L, dtype, device):
    """uninitialised (C, B, L), channel rows row_pitch(B L) elements apart: strides (cs, L, 1)"""
    cs = row_pitch(B * L)
    buf = torch.empty((C, cs), dtype=dtype, device=device)
    return buf[:, :B * L].view(C, B, L) if B * L > 0 else buf[:, :0].view(C, B, L)


def cm_strides(t):
    """(cs, bs) of a channel-major (C, B, L) tensor -- row (c, b) at element c cs + b bs, rows not overlapping -- or None"""
    if t.dim() != 3:
        return None
    C, B, L = t.shape
    if L > 1 and t.stride(2) != 1:
        return None
    bs = t.stride(1) if B > 1 else L
    if bs < L:
        return None
    need = (B - 1) * bs + L
    cs = t.stride(0) if C > 1 else need
    return (cs, bs) if cs >= need else None


def as_cm(t):
    return t if cm_strides(t) is not None else t.contiguous()


def empty_like_cm(t):
    """a new tensor with t's shape and t's (cs, bs) layout"""
    C, B, L = t.shape
    cs, bs = cm_strides(t)
    buf = torch.empty(max(C * cs, 1), dtype=t.dtype, device=t.device)
    return torch.as_strided(buf, (C, B, L), (cs, bs, 1))


def cm_matrix(t):
    """the (C, B L) matrix view of a channel-major tensor whose sequences follow one another inside a channel row (bs = L), else None"""
    C, B, L = t.shape
    cs, bs = cm_strides(t)
    if B > 1 and bs != L:
        return None
    return torch.as_strided(t, (C, B * L), (cs, 1))


def tables_for(device, L):
    """Twiddle tables for sequence length L on `device` (cached per transform size)."""
    M = lib().hyena_fftconv_fft_size(int(L))
    if M == 0:
        raise HyenaLibraryError(f"hyena fftconv: unsupported sequence length L={L} (1 <= L <= {MAX_L})")
    key = (device.index, M, lib().hyena_fftconv_plan(int(L)))
    t = _tables.get(key)
    if t is None:
        with _lock:
            t = _tables.get(key)
            if t is None:
                nbytes = lib().hyena_fftconv_table_bytes(int(L))
                t = torch.empty(nbytes, dtype=torch.uint8, device=device)
                with _backend.guard(device):
                    check(lib().hyena_fftconv_init_tables(t.data_ptr(), int(L), _backend.stream(device)))
                _tables[key] = t
    return t


_retired = []     # (key, buffer): outgrown workspaces a captured hipGraph of that (device, stream) may still replay on: kept alive
_captured = set() # (device index, stream) keys whose workspace was handed out during a hipGraph capture


def workspace_for(device, nbytes):
    """A per-(device, stream) scratch buffer, grown on demand (geometrically) and reused across calls.  An outgrown buffer
    is freed (stream-ordered, through the caching allocator) unless launches captured into a hipGraph point at it -- only
    then is it retired instead, for the life of the process."""
    stream = _backend.stream(device)
    key = (device.index, stream)
    w = _workspace.get(key)
    capturing = getattr(_backend, "capturing", lambda: False)()
    if w is None or w.numel() < nbytes:
        if w is not None:
            if key in _captured:
                _retired.append((key, w))
                _captured.discard(key)
            nbytes = max(int(nbytes), int(1.5 * w.numel()))
        w = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspace[key] = w
    if capturing:
        _captured.add(key)
    return w, stream


def release_stream_state(device, stream):
    """Drop what this module keeps for one (device, stream): its workspace, the record that captured launches point at it, and any
    outgrown buffers retired on its behalf.  Called by the owner of a hipGraph when the graph is released (GraphedTrainStep.release(),
    a sequence-length stage re-capturing): without it every discarded capture stream leaks one workspace."""
    key = (getattr(device, "index", None), stream)
    w = _workspace.pop(key, None)
    _captured.discard(key)
    n = len(_retired)
    _retired[:] = [(k, r) for (k, r) in _retired if k != key]          # the graphs that could replay on them are gone with the stream
    return w is not None or len(_retired) != n


def reset_save_decisions(device=None):
    """Forget the cached keep-the-spectra decisions (all, or one device's): they are taken against the memory that was free when a shape
    was first seen -- re-evaluate them once model + optimizer state exist, at a new GraphedTrainStep, or at a new sequence-length stage."""
    if device is None:
        _save_decision.clear()
        return
    idx = getattr(device, "index", None)
    for key in [k for k in _save_decision if k[0] == idx]:
        del _save_decision[key]


def _chunk_override():
    v = os.environ.get("HYENA_FFTCONV_CHUNK")
    return int(v) if v else 0


def saved_bytes(B, D, L):
    """Size of the optional saved-spectrum buffer (column-transformed filter + activations) for (B, D, L)."""
    return int(lib().hyena_fftconv_saved_bytes(int(B), int(D), int(L)))


_save_decision = {}   # (device index, B, D, L) -> bool


def save_spectra_default(B, D, L, device=None):
    """Keep the forward's column spectra for the backward?  HYENA_FFTCONV_SAVE_SPECTRA = 0 | 1 | auto (default:
    on while the buffer stays below HYENA_FFTCONV_SAVE_LIMIT_GB, default 6 GiB per call)."""
    mode = os.environ.get("HYENA_FFTCONV_SAVE_SPECTRA", "auto").lower()
    if mode in ("0", "off", "false"):
        return False
    if lib().hyena_fftconv_plan(int(L)) == PLAN_ONCHIP:
        # the only intermediate is the filter spectrum H [D][M]: small, and keeping it saves the backward one launch
        # (one transform per channel) -- worth it whenever there is a batch to amortise it over
        return B >= 2 and saved_bytes(B, D, L) <= 256 * 2 ** 20
    if mode in ("1", "on", "true"):
        return True
    need = saved_bytes(B, D, L)
    if need > float(os.environ.get("HYENA_FFTCONV_SAVE_LIMIT_GB", "6")) * 2 ** 30:
        return False
    # per call = per layer: never take more than a quarter of what the device can still give (the recomputing backward
    # fits where this buffer would not).  Decided ONCE per (device, B, D, L): the answer must not flip between steps, nor
    # between an eager warm-up and the hipGraph capture that follows it, with the state of the allocator's cache.
    key = (getattr(device, "index", None), int(B), int(D), int(L))
    hit = _save_decision.get(key)
    if hit is None:
        free = _backend.free_memory(device) if device is not None else _backend.free_memory()
        if free is None:                         # capturing and never decided eagerly: keep the spectra, do not cache
            return True
        hit = _save_decision[key] = bool(need <= free // 4)
    return hit


def fftconv_fwd(u, k, bias, chunk=None, save=False, grad=None):
    """u (B, D, L), k (D, L) fp32, bias (D,) fp32 or None -> out like u.  u and k may be packed or pitched rows (ld_of); out gets u's pitch.
    save=True additionally returns the saved-spectrum buffer for fftconv_bwd(..., saved=).
    grad: a backward of the same shape will follow (default: save or autograd's grad mode) -- the workspace is then sized
    for it right away instead of being outgrown at the first backward."""
    _require_gpu(u, "u")
    B, D, L = u.shape
    u, k = as_rows(u), as_rows(k)
    ldx, ldk = ld_of(u), ld_of(k)
    out = empty_like_rows(u)
    if B == 0:                                   # empty batch: nothing to launch (torch.fft returns an empty tensor too)
        return (out, torch.empty(0, dtype=torch.uint8, device=u.device)) if save else out
    chunk = _chunk_override() if chunk is None else int(chunk)
    tables = tables_for(u.device, L)
    nbytes = lib().hyena_fftconv_workspace_bytes(B, D, L, 0, chunk)
    if (save or torch.is_grad_enabled()) if grad is None else grad:
        # a backward of the same shape follows and needs the larger buffer: take it now instead of outgrowing this one
        nbytes = max(nbytes, lib().hyena_fftconv_workspace_bytes(B, D, L, 1, chunk))
    ws, stream = workspace_for(u.device, nbytes)
    bp = bias.data_ptr() if bias is not None else None
    with _backend.guard(u.device):
        saved = torch.empty(saved_bytes(B, D, L), dtype=torch.uint8, device=u.device) if save else None
        check(lib().hyena_fftconv_fwd_ld(u.data_ptr(), k.data_ptr(), bp, out.data_ptr(), B, D, L, ldx, ldk, dtype_code(u.dtype),
                                         tables.data_ptr(), ws.data_ptr(), ws.numel(), chunk,
                                         saved.data_ptr() if save else None, saved.numel() if save else 0, stream))
    return (out, saved) if save else out


def _same_pitch(t, ld):
    """t with row pitch ld (a copy if it has another one)"""
    if ld_of(t) == ld:
        return t
    r = empty_rows(t.shape[:-1], t.shape[-1], t.dtype, t.device, pitch=ld)
    r.copy_(t)
    return r


def fftconv_bwd(dout, u, k, bias, need_du=True, need_dk=True, chunk=None, saved=None):
    """Returns (du like dout | None, dk (D, L) fp32 | None, dbias (D,) fp32 | None).  Packed or pitched rows as in fftconv_fwd: du gets
    dout's pitch, dk gets k's (row_pitch(L) when k is not given).
    With saved= (from fftconv_fwd(save=True)) k is not read, and u only on the workspace-free path (L <= 32768, where
    the saved buffer holds the filter spectrum alone)."""
    _require_gpu(dout, "dout")
    B, D, L = dout.shape
    dout = as_rows(dout)
    ldx = ld_of(dout)
    if u is not None:
        u = _same_pitch(as_rows(u), ldx)
    if k is not None:
        k = as_rows(k)
    ldk = ld_of(k) if k is not None else row_pitch(L)
    du = empty_like_rows(dout) if need_du else None
    dk = empty_rows((D,), L, torch.float32, dout.device, pitch=ldk) if need_dk else None
    dbias = torch.empty((D,), dtype=torch.float32, device=dout.device) if need_dk else None
    if B == 0:                                   # empty batch: the parameter gradients are sums over nothing
        return du, None if dk is None else dk.zero_(), None if dbias is None else dbias.zero_()
    chunk = _chunk_override() if chunk is None else int(chunk)
    tables = tables_for(dout.device, L)
    nbytes = lib().hyena_fftconv_workspace_bytes(B, D, L, 1, chunk)
    ws, stream = workspace_for(dout.device, nbytes)
    bp = bias.data_ptr() if bias is not None else None
    ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    with _backend.guard(dout.device):
        # (with the forward's spectra the library reads neither k nor -- for L > 32768 -- u; k is passed only without them)
        check(lib().hyena_fftconv_bwd_ld(dout.data_ptr(), ptr(u), None if saved is not None else ptr(k), bp, ptr(du), ptr(dk), ptr(dbias),
                                         B, D, L, ldx, ldk, dtype_code(dout.dtype), tables.data_ptr(), ws.data_ptr(), ws.numel(),
                                         chunk, ptr(saved), saved.numel() if saved is not None else 0, stream))
    return du, dk, dbias


# ---- fused mixer shell (include/hyena_mixer.h): short depthwise conv + gates + layout changes -----------------------
def mixer_pre_fwd(x, w, b, L):
    """x (B, Lx, 3D), w (3D, 3) fp32, b (3D,) fp32 -> vg (B, D, L) = short_conv(x)[v] * short_conv(x)[x1]."""
    _require_gpu(x, "x")
    B, Lx, D3 = x.shape
    D = D3 // 3
    vg = torch.empty((B, D, L), dtype=x.dtype, device=x.device)
    with _backend.guard(x.device):
        check(lib().hyena_mixer_pre_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), vg.data_ptr(), B, L, Lx, D,
                                        dtype_code(x.dtype), _backend.stream(x.device)))
    return vg


def mixer_post_fwd(y, x, w, b):
    """y (B, D, L), x (B, Lx, 3D) -> z (B, L, D) = y^T * short_conv(x)[x0]."""
    _require_gpu(x, "x")
    B, D, L = y.shape
    z = torch.empty((B, L, D), dtype=x.dtype, device=x.device)
    with _backend.guard(x.device):
        check(lib().hyena_mixer_post_fwd(y.data_ptr(), x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), B, L,
                                         x.shape[1], D, dtype_code(x.dtype), _backend.stream(x.device)))
    return z


def mixer_partials(x, L):
    B, Lx, D3 = x.shape
    n = lib().hyena_mixer_partial_floats(B, L, D3 // 3)
    return torch.empty(n, dtype=torch.float32, device=x.device).view(B, -1, D3, 4)


def mixer_post_bwd(dz, y, x, w, b, dx, part):
    """-> dy (B, D, L); fills dx[..., 0:D] (positions < L) and part[..., 0:D, :]."""
    B, D, L = y.shape
    dy = torch.empty_like(y)
    with _backend.guard(x.device):
        check(lib().hyena_mixer_post_bwd(dz.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                         dy.data_ptr(), dx.data_ptr(), part.data_ptr(), B, L, x.shape[1], D,
                                         dtype_code(x.dtype), _backend.stream(x.device)))
    return dy


def mixer_pre_bwd(dvg, x, w, b, dx, part):
    """fills dx[..., D:3D] (positions < L) and part[..., D:3D, :]."""
    B, D, L = dvg.shape
    with _backend.guard(x.device):
        check(lib().hyena_mixer_pre_bwd(dvg.data_ptr(), x.data_ptr(), w.data_ptr(), b.data_ptr(), dx.data_ptr(),
                                        part.data_ptr(), B, L, x.shape[1], D, dtype_code(x.dtype),
                                        _backend.stream(x.device)))


# ---- the shell in channel-major layout (include/hyena_mixer.h, hyena_cm_*) ------------------------------------------------
# xT / dxT / zT / dzT: any (cs, bs) layout (cm_strides); vg / y / dy / dvg: packed or pitched rows (ld_of).  What these functions allocate:
# empty_cm for the first kind, empty_rows for the second.
def cm_pre_fwd(xT, bin_, w, b, L):
    """xT (3D, B, Lx) [in_proj output without bias], bin_ (3D,) fp32 or None -> vg (B, D, L)."""
    _require_gpu(xT, "xT")
    D3, B, Lx = xT.shape
    D = D3 // 3
    xT = as_cm(xT)
    csx, bsx = cm_strides(xT)
    vg = empty_rows((B, D), L, xT.dtype, xT.device)
    with _backend.guard(xT.device):
        check(lib().hyena_cm_pre_fwd_ld(xT.data_ptr(), None if bin_ is None else bin_.data_ptr(), w.data_ptr(), b.data_ptr(), vg.data_ptr(),
                                        B, L, Lx, D, csx, bsx, ld_of(vg), dtype_code(xT.dtype), _backend.stream(xT.device)))
    return vg


def rows_as_cm_strides(t):
    """(cs, bs) that address a (B, D, L) tensor of pitched rows as rows (d, b): cs = its row pitch, bs = D times that (include/hyena_mixer.h)"""
    ld = ld_of(t)
    return ld, t.shape[1] * ld


def cm_post_fwd(y, xT, bin_, w, b, rows_out=False):
    """y (B, D, L), xT (3D, B, Lx) -> zT (D, B, L); rows_out: the same values as a (B, D, L) tensor of pitched rows (the next convolution's input
    when the operator's order is >= 3)."""
    B, D, L = y.shape
    xT, y = as_cm(xT), as_rows(y)
    csx, bsx = cm_strides(xT)
    if rows_out:
        zT = empty_like_rows(y, dtype=xT.dtype)
        csz, bsz = rows_as_cm_strides(zT)
    else:
        zT = empty_cm(D, B, L, xT.dtype, xT.device)
        csz, bsz = cm_strides(zT)
    with _backend.guard(xT.device):
        check(lib().hyena_cm_post_fwd_ld(y.data_ptr(), xT.data_ptr(), None if bin_ is None else bin_.data_ptr(), w.data_ptr(), b.data_ptr(),
                                         zT.data_ptr(), B, L, xT.shape[2], D, csx, bsx, csz, bsz, ld_of(y), dtype_code(xT.dtype),
                                         _backend.stream(xT.device)))
    return zT


def cm_partials(xT, L):
    D3, B, Lx = xT.shape
    n = lib().hyena_cm_partial_floats(B, L, D3 // 3)
    return torch.empty(n, dtype=torch.float32, device=xT.device).view(D3, -1, 8)


def cm_post_bwd(dzT, y, xT, bin_, w, b, dxT, part, dz_rows=False):
    """-> dy (B, D, L) with y's row pitch; fills dxT[0:D] (positions < L) and part[0:D].  dxT must have xT's layout.
    dz_rows: dzT is a (B, D, L) tensor of pitched rows (the gradient a convolution hands back for its input) instead of channel-major (D, B, L)."""
    B, D, L = y.shape
    xT, y = as_cm(xT), as_rows(y)
    csx, bsx = cm_strides(xT)
    if dz_rows:
        dzT = as_rows(dzT)
        assert tuple(dzT.shape) == (B, D, L)
        csz, bsz = rows_as_cm_strides(dzT)
    else:
        dzT = as_cm(dzT)
        csz, bsz = cm_strides(dzT)
    assert cm_strides(dxT) == (csx, bsx)
    dy = empty_like_rows(y)
    with _backend.guard(xT.device):
        check(lib().hyena_cm_post_bwd_ld(dzT.data_ptr(), y.data_ptr(), xT.data_ptr(), None if bin_ is None else bin_.data_ptr(), w.data_ptr(),
                                         b.data_ptr(), dy.data_ptr(), dxT.data_ptr(), part.data_ptr(), B, L, xT.shape[2], D,
                                         csx, bsx, csz, bsz, ld_of(y), dtype_code(xT.dtype), _backend.stream(xT.device)))
    return dy


def cm_pre_bwd(dvg, xT, bin_, w, b, dxT, part):
    """fills dxT[D:3D] (positions < L) and part[D:3D].  dxT must have xT's layout."""
    B, D, L = dvg.shape
    xT, dvg = as_cm(xT), as_rows(dvg)
    csx, bsx = cm_strides(xT)
    assert cm_strides(dxT) == (csx, bsx)
    with _backend.guard(xT.device):
        check(lib().hyena_cm_pre_bwd_ld(dvg.data_ptr(), xT.data_ptr(), None if bin_ is None else bin_.data_ptr(), w.data_ptr(), b.data_ptr(),
                                        dxT.data_ptr(), part.data_ptr(), B, L, xT.shape[2], D, csx, bsx, ld_of(dvg), dtype_code(xT.dtype),
                                        _backend.stream(xT.device)))


# ---- input projection on the matrix cores with the front of the shell in its epilogue (include/hyena_proj.h) --------------------
def proj_supported(B, Lx, D, dtype):
    code = _DTYPES.get(dtype)
    return code is not None and bool(lib().hyena_proj_supported(int(B), int(Lx), int(D), code)) and row_pitch(B * Lx) < 2 ** 31


def inproj_pre_fwd(u, W, bin_, w, b, L):
    """u (B, Lx, D) 16-bit, W (3D, D) same type, bin_ (3D,) fp32 or None, w (3D, 3) fp32, b (3D,) fp32
    -> xT (3D, B, Lx) = W u^T (no bias; empty_cm layout), vg (B, D, L) = short_conv(xT + bin_)[v] * short_conv(xT + bin_)[x1] (pitched rows)."""
    _require_gpu(u, "u")
    B, Lx, D = u.shape
    assert W.shape == (3 * D, D) and W.dtype == u.dtype and u.is_contiguous() and W.is_contiguous()
    xT = empty_cm(3 * D, B, Lx, u.dtype, u.device)
    csx, bsx = cm_strides(xT)
    vg = empty_rows((B, D), L, u.dtype, u.device)
    with _backend.guard(u.device):
        check(lib().hyena_inproj_pre_fwd_ld(u.data_ptr(), W.data_ptr(), None if bin_ is None else bin_.data_ptr(), w.data_ptr(), b.data_ptr(),
                                            xT.data_ptr(), vg.data_ptr(), B, Lx, int(L), D, csx, bsx, ld_of(vg), dtype_code(u.dtype),
                                            _backend.stream(u.device)))
    return xT, vg


def proj_kernel_generation(family, generation=0):
    """include/hyena_proj.h, hyena_proj_kernel_generation: family 0 = out_proj forward, 1 = in_proj forward; generation 0 queries, 1 / 2 selects
    (process-wide)."""
    return int(lib().hyena_proj_kernel_generation(int(family), int(generation)))


def outproj_supported(B, L, Lx, D, dtype):
    code = _DTYPES.get(dtype)
    return code is not None and code != 0 and Lx >= L and bool(lib().hyena_outproj_supported(int(B), int(L), int(D), code))


def outproj_gate_fwd(y, xT, bin_, w, b, W, bias, want_z):
    """y (B, D, L) conv output, xT (3D, B, Lx), bin_ / w / b as cm_post_fwd, W (D, D) out_proj weight (element type of y), bias (D,) fp32
    [values already rounded to the element type] or None -> out (B, L, D) = (y * x0)^T W^T + bias (packed), zT (D, B, L) = y * x0 (empty_cm
    layout) if want_z else None (bit-identical to cm_post_fwd).  One launch: the gate rides on the operand load of the matrix-core product."""
    out, _, _, _, zT = _outproj(y, xT, bin_, w, b, W, bias, want_z, None)
    return out, zT


def outproj_gate_addnorm_fwd(y, xT, bin_, w, b, W, bias, want_z, residual, ln_w, ln_b, eps):
    """outproj_gate_fwd with the block's residual add + LayerNorm in the kernel's epilogue (include/hyena_proj.h,
    hyena_outproj_gate_addnorm_fwd_ld): residual (B L, D) fp32 or None, ln_w / ln_b (D,) fp32
    -> normed (B, L, D) of y's type, residual' (B L, D) fp32, mean, rstd (B L,), zT or None.  The out_proj output itself is never written."""
    return _outproj(y, xT, bin_, w, b, W, bias, want_z, (residual, ln_w, ln_b, eps))


def _outproj(y, xT, bin_, w, b, W, bias, want_z, norm):
    _require_gpu(y, "y")
    B, D, L = y.shape
    y, xT = as_rows(y), as_cm(xT)
    assert W.shape == (D, D) and W.dtype == y.dtype and xT.dtype == y.dtype and W.is_contiguous()
    dev = y.device
    csx, bsx = cm_strides(xT)
    out = torch.empty((B, L, D), dtype=y.dtype, device=dev)
    zT = empty_cm(D, B, L, y.dtype, dev) if want_z else None
    csz, bsz = cm_strides(zT) if want_z else (0, 0)
    res_out = mean = rstd = residual = ln_w = ln_b = None
    eps = 0.0
    if norm is not None:
        residual, ln_w, ln_b, eps = norm
        assert residual is None or (residual.dtype == torch.float32 and residual.is_contiguous() and residual.numel() == B * L * D)
        res_out = torch.empty((B * L, D), dtype=torch.float32, device=dev)
        mean = torch.empty(B * L, dtype=torch.float32, device=dev)
        rstd = torch.empty(B * L, dtype=torch.float32, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    with _backend.guard(dev):
        check(lib().hyena_outproj_gate_addnorm_fwd_ld(y.data_ptr(), xT.data_ptr(), ptr(bin_), w.data_ptr(), b.data_ptr(), W.data_ptr(), ptr(bias),
                                                      ptr(residual), ptr(ln_w), ptr(ln_b), float(eps), out.data_ptr(), ptr(res_out), ptr(mean),
                                                      ptr(rstd), ptr(zT), B, L, xT.shape[2], D, csx, bsx, csz, bsz, ld_of(y),
                                                      dtype_code(y.dtype), _backend.stream(dev)))
    return out, res_out, mean, rstd, zT


def outproj_dgrad_supported(B, L, D, dtype):
    code = _DTYPES.get(dtype)
    return code is not None and code != 0 and bool(lib().hyena_outproj_dgrad_supported(int(B), int(L), int(D), code))


def outproj_dgrad_gate_bwd(dy2, WoT, y, xT, bin_, w, b, dxT):
    """out_proj's input gradient with cm_post_bwd's work in the kernel's epilogue (include/hyena_proj.h, hyena_outproj_dgrad_gate_bwd_ld):
    dy2 (B L, D) 16-bit, WoT (D, D) = out_proj.weight^T contiguous, y (B, D, L), xT (3D, B, Lx); fills dxT[0:D] (positions < L)
    -> dyc (B, D, L) with y's row pitch, part0 (D, runs, 8) whose [:, :, :5].sum(1) are (dw0, dw1, dw2, db_sc, db_in) of channels [0, D)."""
    _require_gpu(y, "y")
    B, D, L = y.shape
    y, xT = as_rows(y), as_cm(xT)
    assert dy2.shape == (B * L, D) and dy2.is_contiguous() and dy2.dtype == y.dtype and WoT.shape == (D, D) and WoT.is_contiguous()
    csx, bsx = cm_strides(xT)
    assert cm_strides(dxT) == (csx, bsx)
    dyc = empty_like_rows(y)
    part = torch.empty(lib().hyena_outproj_dgrad_partial_floats(B, L, D), dtype=torch.float32, device=y.device).view(D, -1, 8)
    with _backend.guard(y.device):
        check(lib().hyena_outproj_dgrad_gate_bwd_ld(dy2.data_ptr(), WoT.data_ptr(), y.data_ptr(), xT.data_ptr(),
                                                    None if bin_ is None else bin_.data_ptr(), w.data_ptr(), b.data_ptr(), dyc.data_ptr(),
                                                    dxT.data_ptr(), part.data_ptr(), B, L, xT.shape[2], D, csx, bsx, ld_of(y),
                                                    dtype_code(y.dtype), _backend.stream(y.device)))
    return dyc, part


def colsum(x2):
    """x2 (P, N) 16-bit contiguous -> (N,) fp32 = x2.sum(0) (the bias gradient of a linear layer) in one pass at the memory rate;
    falls back to torch's reduction for shapes the kernel does not serve (and for host tensors outside the test double)."""
    P, N = x2.shape
    from . import _gradsum
    got = _gradsum.take(x2)              # the tensor's producer (add_norm_bwd) may have summed its columns already
    if got is not None:
        return got
    code = _DTYPES.get(x2.dtype)
    if code is None or not (x2.is_cuda or _backend.name != "hip") or not x2.is_contiguous() or not lib().hyena_colsum_supported(P, N, code):
        return x2.sum(0, dtype=torch.float32)
    out = torch.empty(N, dtype=torch.float32, device=x2.device)
    part = torch.empty(lib().hyena_colsum_partial_floats(P, N), dtype=torch.float32, device=x2.device)
    with _backend.guard(x2.device):
        check(lib().hyena_colsum(x2.data_ptr(), part.data_ptr(), out.data_ptr(), P, N, code, _backend.stream(x2.device)))
    return out


def mlp_supported(P, K, N, dtype):
    code = _DTYPES.get(dtype)
    return code is not None and bool(lib().hyena_mlp_supported(int(P), int(K), int(N), code))


def mlp_fc1_gelu_fwd(x2, W1, b1):
    """x2 (P, K) 16-bit, W1 (N, K) same type, b1 (N,) fp32 (values already rounded to the element type) or None
    -> a = x2 W1^T + b1, h = gelu_tanh(a), both (P, N)."""
    _require_gpu(x2, "x")
    P, K = x2.shape
    N = W1.shape[0]
    a = torch.empty((P, N), dtype=x2.dtype, device=x2.device)
    h = torch.empty((P, N), dtype=x2.dtype, device=x2.device)
    with _backend.guard(x2.device):
        check(lib().hyena_mlp_fc1_gelu_fwd(x2.data_ptr(), W1.data_ptr(), None if b1 is None else b1.data_ptr(), a.data_ptr(), h.data_ptr(),
                                           P, K, N, dtype_code(x2.dtype), _backend.stream(x2.device)))
    return a, h


def mlp_dh_dgelu_bwd(dy2, W2T, a):
    """dy2 (P, K), W2T (N, K) = fc2.weight^T contiguous, a (P, N) -> da (P, N) = (dy2 W2) * gelu_tanh'(a), db1 (N,) fp32 = da.sum(0)."""
    _require_gpu(dy2, "dy")
    P, K = dy2.shape
    N = W2T.shape[0]
    da = torch.empty((P, N), dtype=dy2.dtype, device=dy2.device)
    part = torch.empty(lib().hyena_mlp_partial_floats(P, N), dtype=torch.float32, device=dy2.device)
    with _backend.guard(dy2.device):
        check(lib().hyena_mlp_dh_dgelu_bwd(dy2.data_ptr(), W2T.data_ptr(), a.data_ptr(), da.data_ptr(), part.data_ptr(), P, K, N,
                                           dtype_code(dy2.dtype), _backend.stream(dy2.device)))
    return da, part.view(-1, N).sum(0)                          # fixed-order two-stage sum: deterministic


# ---- fused implicit filter (include/hyena_filter.h) ---------------------------------------------------------------------
def filter_supported(L, E, order, D):
    return bool(lib().hyena_filter_supported(int(L), int(E), int(order), int(D)))


def _filter_params(z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate):
    L, E = z.shape
    p = FilterParams()
    for name, ten in (("z", z), ("t", t), ("w0", w0), ("b0", b0), ("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("w3", w3),
                      ("freq", freq), ("deltas", deltas)):
        if ten is not None:
            _require_gpu(ten, name)
            assert ten.dtype == torch.float32 and ten.is_contiguous(), name
        setattr(p, name, None if ten is None else ten.data_ptr())
    p.shift, p.modulate, p.z_stride, p.L, p.E, p.D = float(shift), int(bool(modulate)), z.stride(0), L, E, w3.shape[0]
    return p


def filter_fwd(z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate, save, compute_dtype=None):
    """z (L, E), t (L,), weights as in include/hyena_filter.h -> k (D, L) fp32, pitched rows [, saved pre-activations].  ``compute_dtype``
    bfloat16 / float16: the graph the reference computes under torch.autocast of that type (hyena_filter16_fwd; the pre-activations are
    kept as (3, 32, P) 32-bit words holding 16-bit pairs, P = the library's row pitch for L); None: the fp32 graph ((3, 64, P) fp32)."""
    p = _filter_params(z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate)
    k = empty_rows((p.D,), p.L, torch.float32, z.device)
    P = int(lib().hyena_filter_row_pitch(p.L))
    if compute_dtype is None:
        saved = torch.empty((3, 64, P), dtype=torch.float32, device=z.device) if save else None
    else:
        saved = torch.empty((3, 32, P), dtype=torch.int32, device=z.device) if save else None
    with _backend.guard(z.device):
        sp = None if saved is None else saved.data_ptr()
        if compute_dtype is None:
            check(lib().hyena_filter_fwd_ld(ctypes.byref(p), k.data_ptr(), ld_of(k), sp, _backend.stream(z.device)))
        else:
            check(lib().hyena_filter16_fwd_ld(ctypes.byref(p), dtype_code(compute_dtype), k.data_ptr(), ld_of(k), sp,
                                              _backend.stream(z.device)))
    return (k, saved) if save else k


def filter_bwd(dk, saved, z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate, need_dz, compute_dtype=None):
    """-> (dw0, db0, dw1, db1, dw2, db2, dw3, dfreq, dz or None); dz is (L, E).  dk (D, L) fp32, packed or pitched rows.  ``compute_dtype`` as
    in filter_fwd (``saved`` must come from the forward of the same type)."""
    p = _filter_params(z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate)
    _require_gpu(dk, "dk")
    assert dk.dtype == torch.float32 and dk.shape == (p.D, p.L)
    dk = as_rows(dk)
    outs = [torch.empty_like(x) for x in (w0, b0, w1, b1, w2, b2, w3, freq)]
    dzt = torch.empty((p.E, p.L), dtype=torch.float32, device=z.device) if need_dz else None
    g = FilterGrads()
    for name, ten in zip(("dw0", "db0", "dw1", "db1", "dw2", "db2", "dw3", "dfreq"), outs):
        setattr(g, name, ten.data_ptr())
    g.dz = None if dzt is None else dzt.data_ptr()
    nbytes = lib().hyena_filter_workspace_bytes(p.L, p.D)
    with _backend.guard(z.device):
        ws, stream = workspace_for(z.device, nbytes)
        if compute_dtype is None:
            assert saved.dtype == torch.float32
            check(lib().hyena_filter_bwd_ld(ctypes.byref(p), dk.data_ptr(), ld_of(dk), saved.data_ptr(), ctypes.byref(g), ws.data_ptr(),
                                            ws.numel(), stream))
        else:
            assert saved.dtype == torch.int32
            check(lib().hyena_filter16_bwd_ld(ctypes.byref(p), dtype_code(compute_dtype), dk.data_ptr(), ld_of(dk), saved.data_ptr(),
                                              ctypes.byref(g), ws.data_ptr(), ws.numel(), stream))
    return tuple(outs) + (None if dzt is None else dzt.t().contiguous(),)


# ---- fused residual add + LayerNorm of a block (include/hyena_block.h) ------------------------------------------------
def add_norm_supported(D, x_dtype, out_dtype):
    try:
        return bool(lib().hyena_add_norm_supported(int(D), dtype_code(x_dtype), dtype_code(out_dtype)))
    except TypeError:
        return False


def add_norm_fwd(x0, residual, weight, bias, eps, out_dtype, dropout_p=0.0, seed=None):
    """x0 (rows, D), residual (rows, D) fp32 or None -> out (rows, D) out_dtype, residual' fp32, mean, rstd (rows,).
    ``dropout_p`` > 0: dropout(x0) inside the pass, decided per element from ``seed`` (a one-element int64 tensor on x0's device)."""
    _require_gpu(x0, "x0")
    if dropout_p > 0.0:
        assert seed is not None and seed.dtype == torch.int64 and seed.numel() == 1 and seed.device == x0.device
    rows, D = x0.shape
    out = torch.empty((rows, D), dtype=out_dtype, device=x0.device)
    res_out = torch.empty((rows, D), dtype=torch.float32, device=x0.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x0.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x0.device)
    if rows == 0:
        return out, res_out, mean, rstd
    with _backend.guard(x0.device):
        check(lib().hyena_dropout_add_norm_fwd(x0.data_ptr(), dtype_code(x0.dtype), None if residual is None else residual.data_ptr(),
                                               weight.data_ptr(), bias.data_ptr(), float(eps), float(dropout_p),
                                               seed.data_ptr() if dropout_p > 0.0 else None, out.data_ptr(), dtype_code(out_dtype),
                                               res_out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, _backend.stream(x0.device)))
    return out, res_out, mean, rstd


def embed_add_norm_supported(V, D, out_dtype):
    try:
        return bool(lib().hyena_embed_add_norm_supported(int(V), int(D), dtype_code(out_dtype)))
    except TypeError:
        return False


def embed_add_norm_fwd(ids, table, weight, bias, eps, out_dtype, dropout_p=0.0, seed=None):
    """ids (rows,) int64, table (V, D) fp32 -> out (rows, D) out_dtype = LayerNorm(dropout(table[ids])), residual' = dropout(table[ids]) fp32,
    mean, rstd (rows,): the embedding gathered inside the first block's add + LayerNorm pass (include/hyena_block.h)."""
    _require_gpu(table, "table")
    rows, (V, D) = ids.numel(), table.shape
    dev = table.device
    out = torch.empty((rows, D), dtype=out_dtype, device=dev)
    res_out = torch.empty((rows, D), dtype=torch.float32, device=dev)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    if rows == 0:
        return out, res_out, mean, rstd
    with _backend.guard(dev):
        check(lib().hyena_embed_add_norm_fwd(ids.data_ptr(), table.data_ptr(), V, weight.data_ptr(), bias.data_ptr(), float(eps),
                                             float(dropout_p), seed.data_ptr() if dropout_p > 0.0 else None, out.data_ptr(),
                                             dtype_code(out_dtype), res_out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D,
                                             _backend.stream(dev)))
    return out, res_out, mean, rstd


def embed_add_norm_bwd(dout, d_res_out, res_out, ids, V, weight, mean, rstd, dropout_p=0.0, seed=None):
    """-> d_table (V, D) fp32 (per-token-class sums of the gradient of the gathered rows, fixed order), dweight (D,), dbias (D,)."""
    _require_gpu(dout, "dout")
    rows, D = dout.shape
    dev = dout.device
    dt = torch.zeros((V, D), dtype=torch.float32, device=dev)
    dw = torch.zeros(D, dtype=torch.float32, device=dev)
    db = torch.zeros(D, dtype=torch.float32, device=dev)
    if rows == 0:
        return dt, dw, db
    part = torch.empty(lib().hyena_embed_add_norm_partial_floats(rows, D), dtype=torch.float32, device=dev)
    with _backend.guard(dev):
        check(lib().hyena_embed_add_norm_bwd(dout.data_ptr(), dtype_code(dout.dtype), None if d_res_out is None else d_res_out.data_ptr(),
                                             res_out.data_ptr(), ids.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             float(dropout_p), seed.data_ptr() if dropout_p > 0.0 else None, dt.data_ptr(), V,
                                             dw.data_ptr(), db.data_ptr(), part.data_ptr(), rows, D, _backend.stream(dev)))
    return dt, dw, db


def add_norm_bwd(dout, d_res_out, res_out, weight, mean, rstd, dx_dtype, need_dres, dropout_p=0.0, seed=None, offer_colsum=True):
    """-> dx0 (rows, D) dx_dtype, d_residual_in fp32 or None, dweight (D,), dbias (D,).  (``dropout_p``, ``seed``): the forward's.
    offer_colsum (16-bit dx0): the kernel also sums dx0's columns -- the bias gradient of the linear layer in front of this norm -- and leaves them
    in the side table of ``_gradsum`` for that layer's backward (``colsum`` below asks there first)."""
    _require_gpu(dout, "dout")
    rows, D = dout.shape
    dev = dout.device
    dx = torch.empty((rows, D), dtype=dx_dtype, device=dev)
    dres = torch.empty((rows, D), dtype=torch.float32, device=dev) if need_dres else None
    if rows == 0:
        return dx, dres, torch.zeros(D, dtype=torch.float32, device=dev), torch.zeros(D, dtype=torch.float32, device=dev)
    dw = torch.empty(D, dtype=torch.float32, device=dev)          # (written in full by the fixed-order reduction: no zero fill -- 32 launches per model step)
    db = torch.empty(D, dtype=torch.float32, device=dev)
    part = torch.empty(lib().hyena_add_norm_partial_floats(rows, D), dtype=torch.float32, device=dev)
    from . import _gradsum
    want = offer_colsum and _gradsum.ENABLED and dx_dtype in (torch.bfloat16, torch.float16)
    cs = torch.empty(D, dtype=torch.float32, device=dev) if want else None
    with _backend.guard(dev):
        check(lib().hyena_dropout_add_norm_bwd_colsum(dout.data_ptr(), dtype_code(dout.dtype), None if d_res_out is None else d_res_out.data_ptr(),
                                                      res_out.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), float(dropout_p),
                                                      seed.data_ptr() if dropout_p > 0.0 else None, dx.data_ptr(), dtype_code(dx_dtype),
                                                      None if dres is None else dres.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                      None if cs is None else cs.data_ptr(), part.data_ptr(), rows, D, _backend.stream(dev)))
    if want:
        _gradsum.offer(dx, cs)
    return dx, dres, dw, db
