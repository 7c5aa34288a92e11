"""`importance` of a model over B particles whose last site is a plate of n datapoints, deferred (one launch over B x n
elements after the program) against the loop form (B lanes walk n elements): milliseconds per call.
  python tools/experiments/deferred_plate_cost.py [B] [n ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import parity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for n in [int(float(a)) for a in (sys.argv[2:] or ["1e4", "1e5", "1e6"])]:
    t_def, t_loop = parity.check_deferred_plate(B=B, n=n, seed=7, timing=True)
    print(json.dumps({"particles": B, "plate_elements": n, "ms_importance_deferred": 1e3 * t_def, "ms_importance_loop_form": 1e3 * t_loop,
                      "speedup": t_loop / t_def, "bit_identical_and_equal_to_the_oracle": True}), flush=True)
