"""bls-bn254_amd -- MI355X-native batched BLS-BN254 verification (host-side Python mirror).

The directory name carries a hyphen (it mirrors the reference repository's name), so it is loaded
through `blsbn254_loader.load()` at the repo root, which registers it as the module `bls_bn254_amd`.

Layout:
  csrc/        HIP kernels (k_*.hip) + C-ABI host code (host.hip, multi.hip), device arithmetic headers
  tools/       constant generator (bn254_consts.h)
  build.py     hipcc driver (builds libblsbn254_hip.so in-tree)
  engine.py    ctypes binding of include/blsbn254.h with the reference's operator names
  sharded.py   one-process-per-GPU sharding over torch.distributed (RCCL)
"""
from .engine import (Bn254Error, Engine, MultiEngine, PreparedKeys, InvalidG1Bytes, InvalidG2Bytes, InvalidGtBytes,  # noqa: F401
                     InvalidScalarBytes, DEFAULT_DST, POP_DST, library_path, load_library)
