"""GB/s of the host-pointer ABI (stenos_compress_generic / stenos_decompress_generic) on the headline data, PCIe included.
usage: python tools/host_rate.py [MiB=1024] [T=4]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(1)
sample = (rng.integers(0, 1 << 12, size=(mib << 20) // 4 + 1000, dtype=np.uint32)).view(np.uint8)
print(bench.host_pointer_rate(T, sample))
