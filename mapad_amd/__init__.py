"""mapad_amd — MI355X-native implementation of the mapAD read-mapping hot path (`mapad map`).

The product is libmapad_amd.so (hand-written HIP kernels for gfx950 behind the C ABI of include/mapad_amd.h).
This package is the thin host-side mirror used by the tests, bench.py and the multi-GPU driver.
"""
from .binding import (BatchResult, Context, Index, MapadError, Params, hits_to_records, lib, make_params, params_from_cli)  # noqa: F401

__all__ = ["BatchResult", "Context", "Index", "MapadError", "Params", "hits_to_records", "lib", "make_params", "params_from_cli"]
