"""Importable alias of the ``super-resolution_amd`` package (its directory name is
not a Python identifier): ``import super_resolution_amd as sra``."""
import importlib
import sys

_pkg = importlib.import_module("super-resolution_amd")
sys.modules[__name__] = _pkg
