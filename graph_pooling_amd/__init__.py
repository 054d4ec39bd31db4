"""graph_pooling_amd — MI355X-native DiffPool forward/backward behind the reference's nn.Module surface.

    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder, GcnEncoderGraph, GcnSet2SetEncoder
    from graph_pooling_amd.set2set import Set2Set

The arithmetic lives in libdiffpool_hip.so (graph_pooling_amd/csrc, C ABI in include/diffpool_hip.h).
Importing this package does not load the library; the first forward does, and raises if it is missing.
"""
__version__ = "0.1.0"
