"""gie — host-side Python mirror of the GIE-mapping per-frame update over the MI355X C-ABI."""
from ._capi import CamParam, Config, CostMapHdr, FrameStats, MultiScanParam, ScanParam, Voxel  # noqa: F401
from .mapper import (LIB_PATH, Mapper, MapperBase, flt2grids_sq, load_library, make_config)  # noqa: F401
from . import scenes, tiling  # noqa: F401
