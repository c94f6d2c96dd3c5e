"""cutseq_amd -- MI355X-native adapter-trimming engine behind cutseq's scheme/CLI surface.

Only what the hot path needs lives here: the scheme parser and presets (``common``), the
op-chain compiler (``plan``), the ctypes binding of the HIP library (``capi``), the batch
engine and the host-side record logic.  There is no CPU trimming path in this package:
without the HIP library and a gfx950 device the engine raises.
"""
from .common import BUILDIN_ADAPTERS, BarcodeConfig, BarcodeSeq, reverse_complement  # noqa: F401

__version__ = "0.5.0"  # the one version source: pyproject.toml reads it (tool.setuptools.dynamic), `cutseq -V` prints it
