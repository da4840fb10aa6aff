"""zigz_amd -- MI355X (gfx950) backend for zigz's sumcheck / MLE / Lasso / SHA3-Merkle hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/zigz_hip.h, and the C++ mirror of
the Zig host), and thin ctypes faces over both.  Attribute access loads the in-tree shared objects and
fails loudly (ImportError) if they are missing; there is no CPU fallback.  `zigz_amd.build` stays
importable without the libraries (it is what produces them)."""
import importlib

_HIP_NAMES = {"P", "NUM_COLUMNS", "CommitJob", "CommitmentScheme", "Context", "SimpleMerkleTree", "Transcript",
              "device_count", "sha256", "sha3_256"}


def __getattr__(name):
    if name in _HIP_NAMES:
        return getattr(importlib.import_module(".hip", __name__), name)
    if name == "ZigzError":
        return importlib.import_module(".errors", __name__).ZigzError
    if name in ("errors", "hip", "host", "build", "_ffi", "shard"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module 'zigz_amd' has no attribute {name!r}")
