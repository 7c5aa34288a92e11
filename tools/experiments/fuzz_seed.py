"""one seed of tests/fuzz_models.py with its spec printed: python tools/experiments/fuzz_seed.py SEED [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import fuzz_models as F
seed = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 7
print([{k: v for k, v in st.items() if k in ("kind", "dist", "n", "T", "flag", "bern", "two", "src")} for st in F.random_spec(np.random.default_rng(seed))])
F.run_one(seed, B=B, verbose=True)
print("ok")
