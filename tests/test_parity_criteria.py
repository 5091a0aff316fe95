"""CPU validation of the parity criteria themselves (tests/parity.py): a second, differently rounded float32 build of the
oracle (FMA contraction) stands in for "another correct float32 implementation" and must PASS; a float32 build with a
deliberate 5 % error in the constraint damping must FAIL.  This is what makes a green GPU ladder mean something."""
import numpy as np
import pytest

from tests import parity


@pytest.fixture(scope="module")
def inputs(oracle_built):
    return parity.rollout_inputs("rodent_optimized", 8, 120, (8, 8), seed=21)


def test_a_differently_rounded_float32_build_passes(inputs):
    seq, A, tab = inputs
    out = parity.substep_ladder(parity.OracleImpl("rodent_optimized", 8, "f32fma", (8, 8)), seq, A,
                                parity.OracleImpl("rodent_optimized", 8, "f32", (8, 8)))
    print(out)
    parity.assert_substep_criteria(out)


def test_a_five_percent_modelling_error_is_rejected(inputs):
    seq, A, tab = inputs
    out = parity.substep_ladder(parity.OracleImpl("rodent_optimized", 8, "f32bug", (8, 8)), seq, A,
                                parity.OracleImpl("rodent_optimized", 8, "f32", (8, 8)))
    print(out["quantiles"])
    with pytest.raises(AssertionError):
        parity.assert_substep_criteria(out)


@pytest.fixture(scope="module")
def newton_inputs(oracle_built):
    return parity.rollout_inputs("rodent_optimized", 8, 120, (4, 8), seed=23, solver="newton")


def test_newton_criteria_pass_for_a_differently_rounded_build_and_reject_the_bug(newton_inputs):
    seq, A, tab = newton_inputs
    gap = parity.OracleImpl("rodent_optimized", 8, "f32", (4, 8), solver="newton")
    out = parity.substep_ladder(parity.OracleImpl("rodent_optimized", 8, "f32fma", (4, 8), solver="newton"), seq, A, gap)
    print(out)
    assert out["niter_equal"] < 0.5                     # why C2 is taken against the float32 oracle for Newton (see assert_substep_criteria)
    parity.assert_substep_criteria(out, newton=True)
    out = parity.substep_ladder(parity.OracleImpl("rodent_optimized", 8, "f32bug", (4, 8), solver="newton"), seq, A, gap)
    with pytest.raises(AssertionError):
        parity.assert_substep_criteria(out, newton=True)
