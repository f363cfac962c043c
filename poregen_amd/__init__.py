"""poregen_amd -- MI355X-native implementation of poregen's `gmove` hot path.

The product is `libpgmove.so` (hand-written HIP kernels for gfx950 behind the C ABI of
include/pgmove.h) plus the `poregen gmove` CLI. This package is the thin Python host layer used by
the tests, bench.py and the multi-GPU launcher: ctypes bindings, a synthetic workload generator and
the rank-sharding logic. There is no CPU fallback: importing `poregen_amd.engine` without the built
library raises.
"""
__version__ = "0.1.0"
