"""What the 1e-4 m / 1e-4 rad claim rests on (scripts/oracle_sensitivity.py): the oracle states evaluation orders the
reference's sources do not spell out, one compile-time switch per statement (oracle/dmsa_oracle.cpp header).  This test builds three of
the alternative oracles and bounds how far the optimised poses move:

  * the two places where the oracle follows the hardware instead of the reference -- fma chain in J^T J for P > 64 (the reference has no
    FMA) and dmsa_detmath.h instead of glibc trigonometry -- must be harmless (< 1e-6), so the product's bit-identity to the oracle
    carries over to a reference with the other choice;
  * a float-order hypothesis (fit sums as float chains) must stay bounded, and is EXPECTED to exceed 1e-4: such orders decide the bar,
    which is why parity with the real reference stays "unpinned" until scripts/build_ref_oracle.sh has run somewhere.
"""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sens():
    spec = importlib.util.spec_from_file_location("oracle_sensitivity", os.path.join(ROOT, "scripts", "oracle_sensitivity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j4", "all", "_variants/libdmsa_oracle_JTJ_NOFMA.so",
                           "_variants/libdmsa_oracle_GLIBC_TRIG.so", "_variants/libdmsa_oracle_FIT_FLOAT.so"])
    return mod


def _deviation(sens, case, hyp, iters):
    import numpy as np

    base = sens.run(case, None, iters)
    v = sens.run(case, os.path.join(ROOT, "oracle", "_variants", f"libdmsa_oracle_{hyp}.so"), iters)
    return max(float(np.abs(np.array(v["gt"]) - np.array(base["gt"])).max()), float(np.abs(np.array(v["go"]) - np.array(base["go"])).max())), \
        v["structure"] == base["structure"]


@pytest.mark.parametrize("case,iters", [("golden_window", 4), ("keyframes_P72", 3), ("keyframes_P186", 2)])
def test_hardware_shaped_statements_are_harmless(sens, case, iters):
    for hyp in ("JTJ_NOFMA", "GLIBC_TRIG"):
        dev, same = _deviation(sens, case, hyp, iters)
        assert dev < 1e-6 and same, (case, hyp, dev)


def test_float_order_statements_decide_the_bar(sens):
    dev, _ = _deviation(sens, "golden_window", "FIT_FLOAT", 4)
    assert 1e-6 < dev < 5e-2, dev  # bounded, but well above what the hardware-shaped statements move
