"""ccs_amd — MI355X-native CCS per-ZMW consensus hot path.

The product is `libccsx.so` (hand-written gfx950 HIP kernels behind the C ABI of include/ccsx.h) and the
`ccs` command-line driver (ccs_amd/bin/ccs).  `ccs_amd.api` is a ctypes mirror of the C ABI used by the
tests and the benchmark; `ccs_amd.shard` holds the ZMW sharder.  There is no CPU fallback.
"""
__version__ = "0.1.0"
