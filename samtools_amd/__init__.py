"""samtools_amd -- MI355X-native mpileup/depth engine behind the samtools command surface.

This package is a thin ctypes binding of the C-ABI library (include/samtools_amd.h).  All the
work happens in samtools_amd/lib/libsamtools_amd.so (hand-written HIP kernels for gfx950 plus the
C++ host drivers).  There is no Python or CPU fallback: importing fails loudly when the library
has not been built, and creating an Engine fails loudly when no HIP device is usable.
"""
from ._capi import (  # noqa: F401
    Engine,
    EngineError,
    KernelTime,
    MplpParams,
    DepthParams,
    GlfParams,
    GlfCol,
    CalmdParams,
    ConsParams,
    ConsCol,
    ConsInfo,
    PlanInfo,
    Reads,
    Window,
    device_count,
    baq_stream_bytes_per_base,
    baq7s_stream_bytes_per_base,
    lib,
    main_depth,
    main_mpileup,
    version,
    MPLP,
    EXPORTED_SYMBOLS,
)

__all__ = [
    "Engine", "EngineError", "KernelTime", "MplpParams", "DepthParams", "GlfParams", "GlfCol", "CalmdParams", "ConsParams", "ConsCol", "ConsInfo", "PlanInfo", "Reads", "Window",
    "device_count", "lib", "main_depth", "main_mpileup", "version", "MPLP", "EXPORTED_SYMBOLS",
]
