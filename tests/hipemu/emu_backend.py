"""Test double for hyena_dna_amd._lib: runs the SAME kernel sources on the CPU under tests/hipemu so that the
host-side logic (ctypes marshalling, autograd wiring, workspace/tables handling, module mirrors) can be tested
without a GPU.  TEST INFRASTRUCTURE ONLY -- installed by the `emu_backend` pytest fixture via monkeypatch."""
import contextlib

from . import build_emu


class EmuBackend:
    name = "hipemu (test only)"

    def __init__(self):
        self.path = build_emu.build()

    def require(self, t, name):
        assert not t.is_cuda

    def guard(self, device):
        return contextlib.nullcontext()

    def stream(self, device):
        return None

    def capturing(self):
        return False

    def free_memory(self, device=None):
        return None
