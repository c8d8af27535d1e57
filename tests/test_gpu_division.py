"""The traversal's short division form (one multiply + four fma on a per-ray refined
reciprocal) must equal the IEEE quotient bit for bit over the operand range the traversal's
static checks admit; the device self test compares the two on 2^32 operand pairs."""
import pytest

from pathtracer_amd import capi

pytestmark = pytest.mark.gpu


def test_short_division_form_equals_ieee_quotient():
    for seed in (1, 0xC0FFEE):
        assert capi.selftest_division(1 << 31, seed=seed) == 0
