"""MI355X-native ESRGAN hot path (RRDBNet generator, spectral-norm discriminator,
VGG-feature loss) behind the reference's nn.Module interface.

The directory name contains a hyphen; import it with
``importlib.import_module("super-resolution_amd")`` or through the
``super_resolution_amd`` alias module at the repository root.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "engine", "esrgan", "utils"]
