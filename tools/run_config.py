#!/usr/bin/env python3
"""Run ONE of BASELINE configs 3 / 4 / 5 exactly as bench.py's `other_configs` does, a fixed number of times and
nothing else — the program rocprofv3 wraps for that config's kernel trace and SQ counters (tools/prof_config.sh):

  python tools/run_config.py <3|4|5> [reps]      prints {"config": .., "units": total workload units run, "programs": .., ...}

`units`: steps (config 3, incl. the sweeps capture() itself runs), runs (4), sweeps (5): tools/make_counters.py divides
the process's total SQ_INSTS_VALU by it.  `programs`: engine.program_digest of the site programs the workload
created — with the library's sha, the identity of the code the counters were taken on (bench.config_valu checks both)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

which = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
from genjax_amd import engine  # noqa: E402

with engine.program_digest() as programs:
    w = bench.config_workload(which)
    for _ in range(reps):
        w["run"]()
torch.cuda.synchronize()
print(json.dumps({"config": which, "unit": w["unit"], "reps": reps,
                  "units": reps * w["units"] + w["warm_units"], "programs": programs.hex()}))
