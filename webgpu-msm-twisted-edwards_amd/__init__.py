"""MI355X-native Twisted-Edwards-BLS12 MSM engine: host-side mirror of the reference's entry point.

The product is `libtemsm.so` (hand-written HIP kernels for gfx950 + the C-ABI of include/te_msm.h).
This package is the thin Python host layer over that C-ABI used by tests, bench.py and
multi-process (one rank per GPU) deployments; the JavaScript host layer that keeps the reference's
`compute_msm(bufferPoints, bufferScalars)` signature lives in js/ (see INTEGRATION.md).

The directory name contains hyphens (it is the reference's name + `_amd`), so import it with
    importlib.import_module("webgpu-msm-twisted-edwards_amd")
There is no CPU fallback anywhere in this package: without libtemsm.so and a HIP device every entry
point raises.
"""
from .binding import (  # noqa: F401
    Bases,
    MsmContext,
    MsmError,
    compute_msm,
    set_bases,
    finalize_host,
    host_tail_features,
    finalize_gathered,
    finalize_sum,
    devices_from_env,
    synth_inputs,
    build_library,
    library_path,
    PARTIAL_BYTES,
    PARTIAL_BYTES_BLS12_377,
    partial_bytes,
    CURVE_TE_BLS12,
    CURVE_BLS12_377_G1,
    WORKSETS,
    MAX_BATCH,
)
from .sharding import ShardedPipeline, compute_msm_sharded, distribute_inputs, exchange_partials, merge_partials, rows_of_batched_msm, window_shard_for_rank  # noqa: F401
