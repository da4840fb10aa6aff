"""zigz_amd -- MI355X (gfx950) backend for zigz's sumcheck / MLE / Lasso / SHA3-Merkle hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/zigz_hip.h, and the C++ mirror of
the Zig host), and thin ctypes faces over both.  Importing this package loads the in-tree shared
objects and fails loudly if they are missing; there is no CPU fallback."""
from . import errors  # noqa: F401
from .errors import ZigzError  # noqa: F401
from .hip import (P, NUM_COLUMNS, CommitJob, CommitmentScheme, Context, SimpleMerkleTree, Transcript,  # noqa: F401
                  device_count, sha256, sha3_256)
