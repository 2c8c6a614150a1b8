"""CPU suite, part 2: the C-ABI library loads, exports every symbol include/pre3.h declares, and fails
loudly (no CPU fallback) when no HIP device is present.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pre3.h")).read()
    return sorted(set(re.findall(r"PRE3_API\s+[\w\s\*]+?\b(pre3_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pre3):
    names = _declared()
    assert len(names) >= 40
    lib = C.CDLL(pre3.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libpre3.so does not export %s" % n
    assert sorted(pre3._lib.EXPORTS) == names


def test_the_test_hooks_are_not_part_of_the_public_header(pre3):
    """pre3_test_stall / pre3_match_shard_test_stall (3pre_amd/csrc/pre3_test_hooks.h) are exported for tests/test_gpu_comm.py, inert without
    PRE3_TEST_HOOKS=1, and not declared in include/pre3.h"""
    txt = open(os.path.join(ROOT, "include", "pre3.h")).read()
    assert "test_stall" not in txt
    lib = C.CDLL(pre3.LIB_PATH)
    assert hasattr(lib, "pre3_test_stall") and hasattr(lib, "pre3_match_shard_test_stall")


def test_no_torch_types_in_header():
    txt = open(os.path.join(ROOT, "include", "pre3.h")).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)          # strip comments
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code and "Tensor" not in code


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "3pre_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "pre3_oracle" not in src, f


@pytest.mark.skipif(os.environ.get("PRE3_EXPECT_GPU") == "1", reason="GPU box")
def test_fails_loudly_without_a_device(pre3):
    if pre3.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pre3.Pre3Error) as e:
        pre3.EkfFilter([250.0, 90, 70, 0, 0, 144, 176], np.zeros(3, np.int32))
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(pre3.Pre3Error):
        pre3.siftmatch(np.zeros((128, 2), np.uint8), np.zeros((128, 2), np.uint8))
    with pytest.raises(pre3.Pre3Error):
        pre3.kNearestNeighbors(np.zeros((4, 2)), np.zeros((1, 2)), 1)
    with pytest.raises(pre3.Pre3Error):
        pre3.update(np.zeros(19), np.eye(19), np.ones((2, 19)) * (np.arange(19) < 13), None, [0.0, 0.0], [0.0, 0.0])


def test_gateway_argument_errors_match_reference_messages(pre3):
    # sift/siftmatch.c:154-190 messages
    with pytest.raises(pre3.Pre3Error, match="same number of rows"):
        pre3.siftmatch(np.zeros((128, 2)), np.zeros((64, 2)))
    with pytest.raises(pre3.Pre3Error, match="same class"):
        pre3.siftmatch(np.zeros((128, 2)), np.zeros((128, 2), np.float32))
    with pytest.raises(pre3.Pre3Error, match="two dimensional"):
        pre3.siftmatch(np.zeros((2, 2, 2)), np.zeros((2, 2)))
    with pytest.raises(pre3.Pre3Error, match="Unsupported numeric class"):
        pre3.siftmatch(np.zeros((4, 2), np.int32), np.zeros((4, 2), np.int32))
