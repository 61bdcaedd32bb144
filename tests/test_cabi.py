"""The product library (hipcc, gfx950) loads on a machine without a GPU and exports every symbol the header
declares; the pure host-side entry points answer without touching a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hyena_fftconv.h")
ALL_HEADERS = sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))


@pytest.fixture(scope="module")
def product_lib():
    from hyena_dna_amd import build
    path = build.build(verbose=False)
    return ctypes.CDLL(path)


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hyena_fftconv_\w+)\s*\(", text)))


def declared_symbols_all_headers():
    out = set()
    for h in ALL_HEADERS:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        out.update(re.findall(r"\b(hyena_\w+)\s*\(", text))        # every entry point of every header: fftconv, mixer, cm, filter, filter16, add_norm, inproj, proj, mlp, colsum ...
    return sorted(out)


def test_every_header_symbol_is_exported(product_lib):
    syms = declared_symbols_all_headers()
    assert {"hyena_mixer_pre_fwd", "hyena_mixer_post_bwd", "hyena_filter_fwd", "hyena_filter_bwd",
            "hyena_filter_supported", "hyena_filter_workspace_bytes", "hyena_filter_saved_bytes", "hyena_add_norm_fwd",
            "hyena_add_norm_bwd", "hyena_add_norm_supported", "hyena_add_norm_partial_floats"} <= set(syms)
    # the families the narrower pattern of round 3 never looked at
    for fam in ("hyena_cm_", "hyena_inproj_", "hyena_mlp_", "hyena_colsum", "hyena_filter16_"):
        assert any(s.startswith(fam) for s in syms), fam
    for s in syms:
        assert hasattr(product_lib, s), s


def test_filter_host_only_entry_points(product_lib):
    L = product_lib
    L.hyena_filter_saved_bytes.restype = ctypes.c_size_t
    L.hyena_filter_workspace_bytes.restype = ctypes.c_size_t
    assert L.hyena_filter_supported(1 << 20, 5, 64, 256) == 1 and L.hyena_filter_supported(1024, 5, 64, 128) == 1
    assert L.hyena_filter_supported(1024, 5, 16, 128) == 0 and L.hyena_filter_supported(1024, 9, 64, 128) == 0
    assert L.hyena_filter_supported(1024, 5, 64, 96) == 0 and L.hyena_filter_supported((1 << 20) + 1, 5, 64, 256) == 0
    # library-owned rows are pitched to 64 words (round 5: the reference trainer's L = max_length - 1 is odd)
    assert L.hyena_filter_row_pitch(1000) == 1024 and L.hyena_filter_row_pitch(1024) == 1024 and L.hyena_filter_row_pitch(1048575) == 1 << 20
    assert L.hyena_filter_saved_bytes(1000) == 3 * 64 * 1024 * 4
    assert L.hyena_filter_workspace_bytes(1000, 256) == (2 * 64 * 1024 + 256 * (256 * 64 + 1024)) * 4
    L.hyena_filter16_saved_bytes.restype = ctypes.c_size_t
    assert L.hyena_filter16_saved_bytes(999) == 3 * 32 * 1024 * 4
    assert L.hyena_filter_fwd_ld(None, None, 0, None, None) == 1 and L.hyena_filter16_bwd_ld(None, 1, None, 0, None, None, None, 0, None) == 1
    assert L.hyena_filter_fwd(None, None, None, None) == 1          # HYENA_ERR_BAD_ARG, no device touched


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("hyena_fftconv_fwd", "hyena_fftconv_bwd", "hyena_fftconv_workspace_bytes", "hyena_fftconv_init_tables",
              "hyena_fftconv_table_bytes", "hyena_fftconv_fft_size", "hyena_fftconv_abi_version",
              "hyena_fftconv_error_string", "hyena_fftconv_default_chunk", "hyena_fftconv_saved_bytes",
              "hyena_fftconv_fwd_save", "hyena_fftconv_bwd_saved", "hyena_fftconv_plan", "hyena_fftconv_fwd_ld", "hyena_fftconv_bwd_ld"):
        assert s in syms


def test_library_exports_every_declared_symbol(product_lib):
    for s in declared_symbols():
        assert hasattr(product_lib, s), s


def test_host_only_entry_points(product_lib):
    L = product_lib
    L.hyena_fftconv_workspace_bytes.restype = ctypes.c_size_t
    L.hyena_fftconv_table_bytes.restype = ctypes.c_size_t
    L.hyena_fftconv_saved_bytes.restype = ctypes.c_size_t
    assert L.hyena_fftconv_saved_bytes(1, 256, 1 << 20) == 2 * 256 * (1 << 20) * 8
    L.hyena_fftconv_error_string.restype = ctypes.c_char_p
    assert L.hyena_fftconv_abi_version() == 3
    assert L.hyena_fftconv_fft_size(1024) == 1024 and L.hyena_fftconv_fft_size(160000) == 163840 and L.hyena_fftconv_fft_size(163841) == 196608
    assert L.hyena_fftconv_fft_size(131072) == 131072 and L.hyena_fftconv_fft_size(131073) == 163840
    assert L.hyena_fftconv_fft_size(450560) == 458752 and L.hyena_fftconv_fft_size(1 << 20) == 1 << 20
    assert L.hyena_fftconv_fft_size(262144) == 262144 and L.hyena_fftconv_fft_size(262145) == 327680
    # L <= 32768: the workspace-free plan, power-of-two sizes
    assert L.hyena_fftconv_fft_size(2049) == 4096 and L.hyena_fftconv_fft_size(4776) == 8192 and L.hyena_fftconv_fft_size(33000) == 65536
    assert L.hyena_fftconv_plan(32768) == 1 and L.hyena_fftconv_plan(32769) == 2 and L.hyena_fftconv_plan(0) == 0
    assert L.hyena_fftconv_fft_size(700000) == 786432 and L.hyena_fftconv_fft_size(786433) == 1 << 20
    assert L.hyena_fftconv_fft_size(458753) == 524288
    assert L.hyena_fftconv_fft_size((1 << 20) + 1) == 0
    assert L.hyena_fftconv_table_bytes(1 << 20) == 4096 * 8
    # workspace = (B + 1 | 2B + 2) * chunk * M * 8 bytes (fwd | bwd)
    assert L.hyena_fftconv_workspace_bytes(1, 256, 1 << 20, 0, 4) == (1 + 1) * 4 * (1 << 20) * 8
    assert L.hyena_fftconv_workspace_bytes(2, 256, 1 << 20, 1, 4) == (2 * 2 + 2) * 4 * (1 << 20) * 8 + 2 * 4 * 2 * 1024 * 8
    c = L.hyena_fftconv_default_chunk(1, 256, 1 << 20, 0)
    assert c == 256
    assert L.hyena_fftconv_default_chunk(8, 128, 1024, 0) == 128
    assert b"workspace" in L.hyena_fftconv_error_string(3)


def test_python_binding_matches_header(product_lib):
    """hyena_dna_amd._lib declares argtypes for exactly the functions the header exports."""
    from hyena_dna_amd import _lib
    src = open(_lib.__file__).read()
    for s in declared_symbols():
        assert ("L." + s) in src, s



def test_block_host_only_entry_points(product_lib):
    """the host-side answers of include/hyena_block.h's round-4 additions (no device touched)"""
    L = product_lib
    L.hyena_embed_add_norm_partial_floats.restype = ctypes.c_size_t
    L.hyena_embed_add_norm_partial_floats.argtypes = [ctypes.c_long, ctypes.c_int]
    L.hyena_add_norm_partial_floats.restype = ctypes.c_size_t
    L.hyena_add_norm_partial_floats.argtypes = [ctypes.c_long, ctypes.c_int]
    BF16, F32 = 1, 0
    assert L.hyena_embed_add_norm_supported(16, 256, BF16) == 1 and L.hyena_embed_add_norm_supported(12, 128, F32) == 1
    assert L.hyena_embed_add_norm_supported(17, 256, BF16) == 0 and L.hyena_embed_add_norm_supported(16, 512, BF16) == 0
    assert L.hyena_embed_add_norm_supported(0, 256, BF16) == 0
    rows, D = 1 << 20, 256
    grid = L.hyena_add_norm_partial_floats(rows, D) // (3 * D)           # (dweight | dbias | round 6: the column sums of dx0) per workgroup
    assert grid == 2048 and L.hyena_embed_add_norm_partial_floats(rows, D) == grid * (2 + 16) * D
    # dropout: p outside [0, 1) or a missing seed is refused before anything is launched
    assert L.hyena_dropout_add_norm_fwd(None, BF16, None, None, None, ctypes.c_float(1e-5), ctypes.c_float(0.1), None, None, BF16, None, None,
                                        None, ctypes.c_long(4), 256, None) == 1
