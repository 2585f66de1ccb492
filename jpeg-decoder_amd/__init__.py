"""MI355X-native baseline-JPEG decode path behind the `zen-jpeg` decoder API.

csrc/      HIP kernels (zj_kernels.hip), the C ABI (zj_api.hip, include/zj.h), host parser (zj_host.cpp)
binding    ctypes declarations of the C ABI
decoder    Decoder / ParsedImage / Batch: the host-side mirror of the reference's API
shard      batch partitioning over GPUs and band splitting of restart-marked images
synth      synthetic JPEG generator for benchmarks and tests
"""
from .decoder import (Batch, ChainError, Context, DecodeError, Decoder, OutputTables, ParsedImage, RangeBatch,  # noqa: F401
                      Workspace, decode_images, decode_noalloc)
