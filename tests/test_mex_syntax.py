"""CPU: the MEX gateways (mex/*.c -- the reference-side binding of INTEGRATION.md) cannot be built without MATLAB's mex.h, but
they can be checked: `gcc -fsyntax-only -Wall -Wextra` against a declarations-only header written from the MEX API documentation
(tests/mex_api_decl/mex.h: it defines and pins nothing) and the real include/pre3.h -- a misspelt ABI entry, a wrong argument count or a
type mismatch against the C ABI fails here."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATEWAYS = sorted(glob.glob(os.path.join(ROOT, "mex", "*.c")))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
@pytest.mark.parametrize("src", GATEWAYS, ids=[os.path.basename(g) for g in GATEWAYS])
def test_gateway_passes_the_compiler_front_end(src):
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
                        "-Werror=int-conversion", "-I", os.path.join(ROOT, "tests", "mex_api_decl"), "-I", os.path.join(ROOT, "include"), src],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "warning" not in r.stderr, r.stderr[-3000:]


def test_there_are_gateways_for_every_drop_in_of_survey_8b():
    names = {os.path.basename(g) for g in GATEWAYS}
    assert {"siftmatch_gateway.c", "update_gateway.c", "predict_gateway.c", "support_gateway.c", "knn_gateway.c", "ekf_ctx_gateway.c", "vo_gateway.c"} <= names
