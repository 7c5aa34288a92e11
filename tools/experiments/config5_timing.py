#!/usr/bin/env python3
"""Times each call of BASELINE config 5's sweep (bench.config_workload(5)) with HIP events, from the first call on:
the check that the rocprofv3 pass over tools/run_config.py (3 calls, no warm-up) and bench.py's timing (one warm call,
then 5) look at the same kernel at the same speed.   python tools/experiments/config5_timing.py [calls]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
from genjax_amd import engine  # noqa: E402

scope = engine.program_digest()
scope.__enter__()
w = bench.config_workload(5)
ms = []
for _ in range(calls):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    w["run"]()
    b.record()
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
scope.__exit__(None, None, None)
print(json.dumps({"config": 5, "ms_per_call": [round(v, 4) for v in ms], "programs": scope.hex()}))
