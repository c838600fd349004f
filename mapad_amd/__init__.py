"""mapad_amd — MI355X-native implementation of the mapAD read-mapping hot path (`mapad map`).

The product is libmapad_amd.so (hand-written HIP kernels for gfx950 behind the C ABI of include/mapad_amd.h).
This package is the thin host-side mirror used by the tests, bench.py and the multi-GPU driver.
"""
import os as _os

# one hardware queue per batch in flight (read by the HIP runtime at its first call; see csrc/mapad_amd.hip: mapad_default_hw_queues)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .binding import (BatchResult, Context, Index, MapadError, Params, hits_to_records, lib, make_params, params_from_cli)  # noqa: F401

__all__ = ["BatchResult", "Context", "Index", "MapadError", "Params", "hits_to_records", "lib", "make_params", "params_from_cli"]
