"""Registers the hyphen-named package directory `bls-bn254_amd/` as the importable module
`bls_bn254_amd` (a hyphen cannot appear in an import statement)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "bls-bn254_amd")


def load():
    if "bls_bn254_amd" in sys.modules:
        return sys.modules["bls_bn254_amd"]
    spec = importlib.util.spec_from_file_location("bls_bn254_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bls_bn254_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
