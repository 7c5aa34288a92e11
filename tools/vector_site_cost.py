"""Cost of a long vector-valued site under ImportanceK against its `vmap`-plate spelling:
`y ~ normal(a * xs + b, 0.5)` with n observations, K particles.  Usage: python tools/vector_site_cost.py [K] > out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genjax_amd import _lib  # noqa: E402

_lib.install(None)
from oracle import genjax_oracle as O  # noqa: E402  (parity.py imports it at module level; only timings are taken here)

O.build()
from tests import parity  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
out = {"K": K, "what": "seconds per ImportanceK.run_smc; vector_site = one site `normal(a * xs + b, 0.5) @ 'y'`, "
                        "vmap_plate = the same likelihood as point.vmap(...)(a, b, xs) @ 'ys'"}
for n in (500, 5000):
    parity.check_long_vector_sites(n=n, K=33, seed=2)          # (bit-exact vs the oracle at a small K first)
    t = parity.time_vector_site_vs_plate(n=n, K=K)
    t["ratio_vector_over_plate"] = t["vector_site"] / t["vmap_plate"]
    out[str(n)] = t
print(json.dumps(out))
