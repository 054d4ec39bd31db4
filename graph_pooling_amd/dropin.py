"""Make the reference's drivers import these modules under the reference's own module names.

    import graph_pooling_amd.dropin as dropin
    dropin.install()          # sys.modules['encoders'|'set2set'|'aggregators'|'graphsage'] -> HIP versions
    import train              # the reference's train.py now builds and trains the MI355X encoders

train.py does `import encoders` and calls encoders.SoftPoolingGcnEncoder / GcnSet2SetEncoder /
GcnEncoderGraph (train.py:493-508): nothing else in it needs to change.
"""
import sys


def install():
    from . import aggregators, encoders, graphsage, set2set
    sys.modules["encoders"] = encoders
    sys.modules["set2set"] = set2set
    sys.modules["aggregators"] = aggregators
    sys.modules["graphsage"] = graphsage
    return encoders
