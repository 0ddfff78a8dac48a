"""GPU parity of the PointCloud2 decoder through include/dmsa_wire_formats.h: byte work, bit-exact."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import wire_formats as wf
from dmsa_lidar_slam_amd.api import DmsaError
from wire_util import LAYOUTS, make_msg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sensor", list(LAYOUTS))
def test_decoder_matches_oracle(orc, sensor):
    dec = wf.PointCloud2Decoder(sensor)
    for n, height in ((131072, 128), (1, 1), (4097, 1)):
        msg, exp = make_msg(sensor, n, seed=n, height=height)
        if sensor == "unknown" and dec.lastPcMsgStamp < 0:
            assert dec.decode(msg) is None  # first message only seeds lastPcMsgStamp (:388-392)
            msg.stamp += 0.1
        got = dec.decode(msg)
        ref = orc.decode_pointcloud2(msg, sensor, msg.stamp - dec.lastPcMsgStamp)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)
        assert np.array_equal(got[0][:, 0], exp["x"]) and not got[0][:, 3].any()
        if "stamp" in exp:
            assert np.array_equal(got[1], exp["stamp"]) and np.array_equal(got[2], exp["id"])
    dec.close()


def test_decoder_rejects_short_messages():
    dec = wf.PointCloud2Decoder("sick")
    msg, _ = make_msg("velodyne", 100)  # 6 fields: "sick" needs fields 8 and 11
    with pytest.raises(DmsaError):
        dec.decode(msg)
    msg, _ = make_msg("sick", 100)
    msg.data = msg.data[:-8]  # blob shorter than n * point_step
    with pytest.raises(DmsaError):
        dec.decode(msg)
    msg, _ = make_msg("sick", 0)
    xyz, st, ids = dec.decode(msg)
    assert xyz.shape == (0, 4) and st.shape == (0,) and ids.shape == (0,)
    dec.close()
