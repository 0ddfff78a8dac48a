"""The f-rows composed: PointCloud2 bytes -> preProcess -> window setup -> addStaticPoints -> optimizeSet -> keyframe creation -> TUM
lines on a synthetic drive (examples/sequence_demo.py, a miniature of DmsaSlam::processPointCloud)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_sequence_demo_tracks_the_motion():
    import sequence_demo

    r = sequence_demo.run(scans=12)
    assert r["windows"] == 8 and r["keyframes"] >= 1
    assert r["max_position_error_m"] < 0.25  # 1 m/s drive, 0.8 s of windows chained through updateInitialGuess
    log = r["log"]
    assert all(e["iterations"] >= 1 and e["gaussians"] >= 30 for e in log)
    assert log[0]["static"] == 0 and all(e["static"] > 500 for e in log[1:])  # the map exists from the second window on
    assert all(0.0 < e["overlap"] <= 1.0 for e in log[1:])
    assert len(r["tum"]) == 8 and all(len(line.split()) == 8 for line in r["tum"])
    stamps = [float(line.split()[0]) for line in r["tum"]]
    assert np.all(np.diff(stamps) > 0.09)
