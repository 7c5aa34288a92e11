"""genjax_amd — MI355X-native implementation of GenJAX's vectorised inference
hot path (simulate / importance / assess for `@gen` static functions and the
`genjax.inference.smc` combinators) behind the reference's API names
(src/genjax/__init__.py:33-41).  All device work goes through
the C-ABI in include/genmi.h to hand-written HIP kernels for gfx950.
"""
from . import numpy, random
from .core.choice_map import (ChoiceMap, ChoiceMapBuilder, ChoiceMapNoValueAtAddress, Selection,
                              SelectionBuilder)
from .core.generative import (Diff, DiffAnnotate, EditRequest, EmptyRequest, GenerativeFunction,
                              GenerativeFunctionClosure, IndexRequest, NoChange, NotSupportedEditRequest, Regenerate,
                              Trace, UnknownChange, Update)
from .core.generative import Argdiffs, Arguments, Retdiff, Score, VectorRequest, Weight
from .core.mask import Indexed, Mask
from .core.choice_map import DynamicIndex, dynamic_index
from .core.pytree import Closure, Const, PythonicPytree, Pytree, R, nth
from .distributions import (Distribution, bernoulli, beta, categorical, dirichlet, exact_density, flip, half_cauchy,
                            exponential, half_normal, log_normal, normal, tfp_distribution, uniform)
from .static import (AddressReuse, MissingAddress, Rejuvenate, StaticGenerativeFunction, StaticRequest,
                     StaticTrace, gen, trace)
from . import inference
from .inference import Target
from .inference import requests, smc          # `genjax.smc`, `genjax.requests` (inference/__init__.py:22-37 star-exported)
from .inference.sp import Algorithm, Marginal, marginal
from .transforms import jit, vmap
from .combinators import (MaskCombinator, RepeatCombinator, Scan, Vmap, accumulate, iterate, iterate_final, mask, masked_iterate,
                          masked_iterate_final, reduce, repeat, scan)

ExactDensity = Distribution
SampleDistribution = Distribution          # sp.py:100-103: distributions whose value is a ChoiceMap
Address = AddressComponent = object        # typing aliases of core/generative (static addresses here)
trace_p = "trace"                          # the reference's jax primitive; `trace(addr, gen_fn, args)` is the entry
key = random.key
split = random.split
fold_in = random.fold_in
from .engine import clear_caches  # noqa: E402  (build addition: drop every cached site program)

__all__ = [
    "numpy", "random", "inference", "smc", "requests", "ChoiceMap", "ChoiceMapBuilder", "Selection", "SelectionBuilder",
    "ChoiceMapNoValueAtAddress", "Diff", "DiffAnnotate", "EditRequest", "EmptyRequest",
    "GenerativeFunction", "GenerativeFunctionClosure", "NoChange", "UnknownChange", "Regenerate",
    "Trace", "Update", "Mask", "Distribution", "ExactDensity", "bernoulli", "beta", "categorical", "dirichlet",
    "flip", "normal", "uniform", "AddressReuse", "MissingAddress", "Rejuvenate",
    "StaticGenerativeFunction", "StaticRequest", "StaticTrace", "gen", "trace", "Target", "jit",
    "vmap", "key", "split", "fold_in", "NotSupportedEditRequest", "Vmap", "repeat", "RepeatCombinator", "Scan", "scan", "IndexRequest",
    "VectorRequest", "Argdiffs", "Arguments", "Retdiff", "Score", "Weight", "Address", "AddressComponent", "R",
    "Closure", "Const", "PythonicPytree", "Pytree", "nth", "exact_density", "tfp_distribution", "half_cauchy",
    "half_normal", "log_normal", "exponential", "Algorithm", "SampleDistribution", "Marginal", "marginal", "trace_p", "clear_caches",
    "iterate", "iterate_final", "accumulate", "reduce", "mask", "MaskCombinator", "masked_iterate", "masked_iterate_final", "Indexed", "DynamicIndex", "dynamic_index",
]
