"""libsafecrypto_amd - MI355X (gfx950) batched lattice-arithmetic engine behind
libsafecrypto's polynomial hot path (NTT / pointwise / CDT sampler).

The product is the C-ABI shared library libsafecrypto_amd/lib/libsc_hip.so (include/sc_hip.h).
This Python package is only the thin host binding used by tests and bench.py; torch is used
for device memory and streams, never for the arithmetic.
"""
from .engine import Engine, NttPlan, CdtSampler, Dilithium, Bliss, Falcon, Tesla, Rlwe, Ens, Dlp, EnsDlpSigner, Kyber, load_library, LIB_PATH, EngineError, build_cdt_table, Multi, shard_range  # noqa: F401
