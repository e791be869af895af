"""The packed forms of MODEL_SPEC's scalar functions (csrc/spec_math.hip.h: two results per VALU instruction, a shorter
clamp, the integer part of exp's argument from a magic-number add, tanh's quotient without the scaling steps of the
general division) against the scalar definitions the oracle pins -- on the device, for every float32 bit pattern."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which,name", [(0, "exp"), (1, "tanh"), (2, "gelu"), (3, "sigmoid")])
def test_packed_function_equals_scalar_definition_for_every_float(bv, product, which, name):
    abi = bv.bind_batch(product)
    first = ctypes.c_uint(0)
    bad = abi.BeatriceHip_MathSelfTest(which, ctypes.byref(first))
    assert bad == 0, "%s: %d of 2^32 inputs differ, first at bits 0x%08x" % (name, bad, first.value)
