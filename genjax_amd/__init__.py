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
from .core.mask import Mask
from .distributions import (Distribution, bernoulli, beta, categorical, dirichlet, flip, normal, uniform)
from .static import (AddressReuse, MissingAddress, Rejuvenate, StaticGenerativeFunction, StaticRequest,
                     StaticTrace, gen, trace)
from . import inference
from .inference import Target
from .transforms import jit, vmap
from .combinators import Scan, Vmap, repeat, scan

ExactDensity = Distribution
key = random.key
split = random.split
fold_in = random.fold_in

__all__ = [
    "numpy", "random", "inference", "ChoiceMap", "ChoiceMapBuilder", "Selection", "SelectionBuilder",
    "ChoiceMapNoValueAtAddress", "Diff", "DiffAnnotate", "EditRequest", "EmptyRequest",
    "GenerativeFunction", "GenerativeFunctionClosure", "NoChange", "UnknownChange", "Regenerate",
    "Trace", "Update", "Mask", "Distribution", "ExactDensity", "bernoulli", "beta", "categorical", "dirichlet",
    "flip", "normal", "uniform", "AddressReuse", "MissingAddress", "Rejuvenate",
    "StaticGenerativeFunction", "StaticRequest", "StaticTrace", "gen", "trace", "Target", "jit",
    "vmap", "key", "split", "fold_in", "NotSupportedEditRequest", "Vmap", "repeat", "Scan", "scan", "IndexRequest",
]
