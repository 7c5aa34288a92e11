# NOTE: needs the multi-step launch of profiles/r04l_multi_step_attempt.patch applied (git apply); the product does not carry it
# (measured slower than one launch per step: profiles/r04l_multi_step_results.json).
import sys, json
sys.path.insert(0, "/root/repo")
from tests import parity
for kw in (dict(n=200_000, T=25, steps_per_launch=10), dict(n=200_000, T=25, steps_per_launch=10, capture=True),
           dict(n=1_000_000, T=12, steps_per_launch=4, capture=True), dict(n=1_048_576, T=8, steps_per_launch=10)):
    r = parity.check_lgssm_sweep(noise_ahead=True, **kw)
    print(json.dumps({**kw, **{k: v for k, v in r.items() if not hasattr(v, "shape")}}), flush=True)
