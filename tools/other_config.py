#!/usr/bin/env python3
"""One entry of bench.py's "other_configs" on its own.  usage: python tools/other_config.py <name> [--no-cpu]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1]
print(json.dumps({name: bench.measure_other(name, torch, dist, "cuda:0", "--no-cpu" not in sys.argv)}), flush=True)
