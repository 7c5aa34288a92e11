"""The reference-held expectations, on the HIP library.

tests/test_host_logic.py translates the reference's own behavioural tests
  tests/generative_functions/test_distributions.py:25-193   (leaf GFI weight rules)
  tests/generative_functions/test_static_gen_fn.py:287-318  (score == assess; the literal -2.837877)
  tests/inference/test_smc.py:50,57,87                      (log-ML tolerances 1e-1 / 1e-3 / 1e-1)
  tests/inference/test_requests.py:38-166                   (update / regenerate / rejuvenate identities, MH convergence)
  README.md:88-123                                          (BASELINE config 1: beta-bernoulli, ImportanceK k = 50 x 50 trials)
and runs them against the CPU mirror of the C-ABI; tests/test_ref_static.py and tests/test_ref_combinators.py hold the
rest of the reference's in-scope behavioural suite (test_static_gen_fn.py, test_scan / test_vmap / test_repeat_combinator.py,
core/generative/test_core.py).  Here the SAME test bodies run through libgenmi_hip.so on an MI355X: the classes are
subclassed unchanged, only the backend fixture differs (the module-level `usefixtures("hostsim")` of the CPU files
does not apply to classes defined here).
"""
import pytest
import torch

from tests import parity
from tests import test_host_logic as H
from tests import test_ref_combinators as RC
from tests import test_ref_mask_combinator as RMC
from tests import test_ref_static as RS

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _on_the_hip_library(gpu, monkeypatch):
    # the translated tests call `.numpy()` on results (CPU tensors under the mirror): bring device tensors back first
    orig = torch.Tensor.numpy
    monkeypatch.setattr(torch.Tensor, "numpy", lambda self, *a, **k: orig(self.detach().cpu(), *a, **k))
    yield gpu


class TestDistributionsOnDevice(H.TestDistributions):
    pass


class TestStaticOnDevice(H.TestStatic):
    pass


class TestSMCOnDevice(H.TestSMC):
    """incl. test_readme_quickstart = BASELINE config 1"""


class TestRequestsOnDevice(H.TestRequests):
    pass


# ---- tests/test_ref_static.py (test_static_gen_fn.py, 41 reference tests) on the HIP library ----
class TestRefStaticMetadataOnDevice(RS.TestMetadata): pass
class TestRefStaticMiscOnDevice(RS.TestMisc): pass
class TestRefStaticSimulateOnDevice(RS.TestSimulate): pass
class TestRefStaticAssessOnDevice(RS.TestAssess): pass
class TestRefStaticCustomPytreeOnDevice(RS.TestCustomPytree): pass
class TestRefStaticImportanceOnDevice(RS.TestImportance): pass
class TestRefStaticUpdateOnDevice(RS.TestUpdate): pass
class TestRefStaticAddressChecksOnDevice(RS.TestAddressChecks): pass
class TestRefStaticClosuresOnDevice(RS.TestForwardRefAndClosures): pass
class TestRefStaticEditRequestOnDevice(RS.TestStaticEditRequest): pass
class TestRefStaticInlineOnDevice(RS.TestInline): pass
class TestRefStaticMethodsOnDevice(RS.TestMethodsAndPartialApply): pass


# ---- tests/test_ref_combinators.py (test_scan / test_vmap / test_repeat_combinator.py, test_core.py) ----
class TestRefIterateSimpleNormalOnDevice(RC.TestIterateSimpleNormal): pass
class TestRefIterateOnDevice(RC.TestIterate): pass
class TestRefAccumulateReduceOnDevice(RC.TestAccumulateReduce): pass
class TestRefScanBehaviourOnDevice(RC.TestScanBehaviour): pass
class TestRefScanEditsOnDevice(RC.TestScanRegenerateAndIndexRequest): pass
class TestRefVmapOnDevice(RC.TestVmap): pass
class TestRefVmapIndexRequestOnDevice(RC.TestVmapIndexRequest): pass
class TestRefRepeatOnDevice(RC.TestRepeat): pass
class TestRefCoreOnDevice(RC.TestCore): pass
class TestRefMaskCombinatorOnDevice(RMC.TestMaskCombinator): pass


def test_nested_marginal_and_change_target_on_device():
    """A12 / F4 (ref smc.py:214-225, 370-396, 432-465; sp.py:229-238) against the oracle"""
    parity.check_nested_marginal()


def test_change_target_away_from_subset_constraints_on_device():
    """A7 / A9 (ref smc.py:370-396, sp.py:89-91, choice_map.py:658-663, 1494-1496, 1714-1743) against the oracle and f64"""
    parity.check_change_target_from_subset_constraints()
    parity.check_change_target_from_subset_constraints(n=40, k=17, seed=9)
    parity.check_change_target_from_subset_constraints(n=24, k=4096, seed=11)


def test_jax_docs_values_through_the_product_on_device():
    H.test_jax_docs_values_through_the_product()


def test_api_surface_and_derived_distributions_on_device():
    H.test_api_surface_and_derived_distributions()
