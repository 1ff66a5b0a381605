"""hacktv_amd -- Python binding of libhvk, the MI355X composite-video -> IQ
engine. The product is the C ABI in include/hacktv_amd.h; this package only
loads it with ctypes for the test-suite, bench.py and __graft_entry__.py."""
from .ctypes_defs import (HvkConfig, HvkInfo, HvkRational, FLAG_FILTER, FLAG_NOAUDIO,
                          FLAG_NONICAM, FLAG_NOCOLOUR, HVK_OK, HVK_ERROR, HVK_OUT_OF_MEMORY, HVK_NO_DEVICE,
                          HVK_UNSUPPORTED, LEVELS_AUTO, LEVELS_TABLE, LEVELS_COMPUTE)
from .engine import Engine, Group, HvkError, lib, preset, LIB_PATH

__all__ = ["Engine", "Group", "HvkError", "HvkConfig", "HvkInfo", "HvkRational", "lib", "preset", "LIB_PATH",
           "FLAG_FILTER", "FLAG_NOAUDIO", "FLAG_NONICAM", "FLAG_NOCOLOUR", "HVK_OK", "HVK_ERROR",
           "HVK_OUT_OF_MEMORY", "HVK_NO_DEVICE", "HVK_UNSUPPORTED", "LEVELS_AUTO", "LEVELS_TABLE", "LEVELS_COMPUTE"]
