"""`genjax.inference.requests` (src/genjax/inference/requests.py): Rejuvenate is
on the hot path; HMC / SafeHMC need reverse-mode gradients of the site program
(SURVEY.md §8f item 3, next tier)."""
from ..static import Rejuvenate


class HMC:
    def __init__(self, *a, **k):
        raise NotImplementedError("HMC: SURVEY.md §8(f) item 3 (next tier)")


SafeHMC = HMC
__all__ = ["Rejuvenate", "HMC", "SafeHMC"]
