"""Tracking-dataset format: ROS bag v2.0 container + the two message types +
ground_truth.txt, without ROS.  The reference ships no bag, so these pin the module's own
writer -> reader round trip, the container invariants a rosbag reader relies on, and the
reference's documented behaviours (topics, leading slash, exact-stamp synchronisation,
admissible stamp difference, frame-0 camera matrix, no overwrite)."""
import os
import struct

import numpy as np
import pytest

from dbot_ros_amd import dataset as ds
from dbot_ros_amd import synth


def _frames(n=6, rows=24, cols=32, seed=0):
    rng = np.random.default_rng(seed)
    K = synth.camera_matrix(cols, rows)
    out = []
    for k in range(n):
        d = (0.5 + rng.random((rows, cols))).astype(np.float32)
        d[rng.random((rows, cols)) < 0.1] = np.nan
        stamp = ds.Stamp.from_sec(1400000000.25 + k / 30.0)
        out.append((ds.Image(d, stamp, seq=k), ds.CameraInfo(K, rows, cols, stamp, seq=k), np.arange(12.0) + k))
    return out


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_store_and_load_round_trip(tmp_path, compression):
    path = tmp_path / "set"
    a = ds.TrackingDataset(path, load=False)
    for im, info, gt in _frames():
        a.add_frame(im, info, ground_truth=gt)
    if compression == "none":
        a.store()
    else:
        os.makedirs(path)
        msgs = []
        for f in a.data:
            msgs += [(a.image_topic, f.image.stamp, f.image), (a.info_topic, f.info.stamp, f.info)]
        ds.write_bag(path / ds.OBSERVATIONS_FILENAME, msgs, compression="bz2", chunk_messages=5)
        with open(path / ds.GROUND_TRUTH_FILENAME, "w") as fh:
            for f in a.data:
                fh.write(str(f.image.stamp) + " " + " ".join(repr(float(x)) for x in f.ground_truth) + "\n")
    assert sorted(os.listdir(path)) == ["ground_truth.txt", "measurements.bag"]
    b = ds.TrackingDataset(path)
    assert b.size() == a.size() == 6
    for fa, fb in zip(a.data, b.data):
        assert fb.image.stamp == fa.image.stamp and fb.info.stamp == fa.info.stamp
        assert np.array_equal(fb.image.depth, fa.image.depth, equal_nan=True)
        assert np.array_equal(fb.info.K, fa.info.K)
        assert np.array_equal(fb.ground_truth, fa.ground_truth)
    assert np.array_equal(b.get_camera_matrix(3), a.data[0].info.K)
    with pytest.raises(FileExistsError):
        b.store()


def test_container_layout(tmp_path):
    """What a stock rosbag reader needs: magic, a 4 096-byte bag-header record whose index_pos
    points at the connection records, chunk sizes, index offsets that land on message records."""
    frames = _frames(4)
    msgs = []
    for im, info, _ in frames:
        msgs += [("/" + ds.IMAGE_TOPIC, im.stamp, im), ("/" + ds.INFO_TOPIC, info.stamp, info)]
    p = tmp_path / "m.bag"
    ds.write_bag(p, msgs, chunk_messages=3)
    buf = open(p, "rb").read()
    assert buf.startswith(b"#ROSBAG V2.0\n")
    recs = list(ds._records(buf, 13))
    h0, d0 = recs[0]
    assert h0["op"] == b"\x03" and 13 + 4 + len(ds._pack_header([(k, v) for k, v in h0.items()])) + 4 + len(d0) == 13 + 4096
    (index_pos,) = struct.unpack("<Q", h0["index_pos"])
    (conn_count,) = struct.unpack("<I", h0["conn_count"])
    (chunk_count,) = struct.unpack("<I", h0["chunk_count"])
    tail = list(ds._records(buf, index_pos))
    assert [h["op"] for h, _ in tail] == [b"\x07"] * conn_count + [b"\x06"] * chunk_count
    assert conn_count == 2 and chunk_count == 3
    # every index entry of every chunk points at a message-data record of its connection
    chunk = None
    for h, d in recs[1:]:
        if h["op"] == b"\x05":
            chunk = d
            assert len(d) == struct.unpack("<I", h["size"])[0]
        elif h["op"] == b"\x04":
            (conn,) = struct.unpack("<I", h["conn"])
            for k in range(struct.unpack("<I", h["count"])[0]):
                secs, nsecs, off = struct.unpack_from("<III", d, 12 * k)
                hh, _ = next(ds._records(chunk, off))
                assert hh["op"] == b"\x02" and struct.unpack("<I", hh["conn"])[0] == conn
                assert struct.unpack("<II", hh["time"]) == (secs, nsecs)
    # chunk infos point at the chunks
    for h, _ in tail[conn_count:]:
        (pos,) = struct.unpack("<Q", h["chunk_pos"])
        hh, _ = next(ds._records(buf, pos))
        assert hh["op"] == b"\x05"
    got = ds.read_bag(p)
    assert [(t, ty) for t, ty, _, _ in got[:2]] == [("/" + ds.IMAGE_TOPIC, "sensor_msgs/Image"),
                                                     ("/" + ds.INFO_TOPIC, "sensor_msgs/CameraInfo")]


def test_synchronisation_and_other_topics(tmp_path):
    """Only image/info pairs with identical header stamps become frames; messages on other
    topics and unmatched messages are dropped; topics may carry a leading slash."""
    frames = _frames(5)
    msgs = []
    for k, (im, info, _) in enumerate(frames):
        if k != 1:
            msgs.append(("/" + ds.IMAGE_TOPIC, im.stamp, im))
        if k != 3:
            msgs.append((ds.INFO_TOPIC, info.stamp, info))
        msgs.append(("some/other/image", im.stamp, im))
    path = tmp_path / "set"
    os.makedirs(path)
    ds.write_bag(path / ds.OBSERVATIONS_FILENAME, msgs)
    d = ds.TrackingDataset(path)
    assert [f.image.seq for f in d.data] == [0, 2, 4]
    assert all(f.ground_truth.size == 0 for f in d.data)      # no ground_truth.txt: still loads


def test_ground_truth_attachment(tmp_path):
    frames = _frames(4)
    path = tmp_path / "set"
    a = ds.TrackingDataset(path, load=False)
    for im, info, _ in frames:
        a.add_frame(im, info)
    a.store()
    t0 = frames[0][0].stamp.to_sec()
    with open(path / ds.GROUND_TRUTH_FILENAME, "w") as fh:
        fh.write(f"{t0 + 0.005:.9f} 1 2 3\n")                       # within 0.02 s of frame 0 only
        fh.write(f"{t0 + 2 / 30.0 - 0.010:.9f} 4 5 6\n")            # frame 2 only (frame 1 is 0.023 s away)
        fh.write(f"{t0 + 3 / 30.0 + 0.05:.9f} 7 8 9\n")             # nobody
    b = ds.TrackingDataset(path)
    assert [list(f.ground_truth) for f in b.data] == [[1, 2, 3], [], [4, 5, 6], []]
    c = ds.TrackingDataset(path, load=False)
    c.load(first_line_only=True)                                      # the reference as written
    assert [list(f.ground_truth) for f in c.data] == [[1, 2, 3], [], [], []]


def test_frame_vector_and_raw_depth():
    rng = np.random.default_rng(1)
    d = rng.random((12, 16)).astype(np.float32)
    im = ds.Image(d, 10.5)
    assert np.array_equal(im.to_vector(), d.ravel())
    assert np.array_equal(im.to_vector(4), d[::4, ::4].ravel())       # eval(r, c) = native(4r, 4c)
    # 16UC1 millimetres with 0 = no reading
    mm = (d * 1000).astype("<u2")
    mm[0, 0] = 0
    raw = (ds._wr_header(0, ds.Stamp(1, 2), "x") + struct.pack("<II", 12, 16) + ds._wr_string("16UC1") +
           struct.pack("<BI", 0, 32) + struct.pack("<I", mm.nbytes) + mm.tobytes())
    back = ds.Image.deserialize(raw)
    assert np.isnan(back.depth[0, 0]) and np.allclose(back.depth.ravel()[1:], mm.ravel()[1:] * 1e-3)
    assert str(ds.Stamp(12, 5)) == "12.000000005" and ds.Stamp.from_sec(12.000000005) == ds.Stamp(12, 5)


def test_malformed_bags_are_rejected(tmp_path):
    p = tmp_path / "x.bag"
    p.write_bytes(b"not a bag")
    with pytest.raises(ds.BagFormatError):
        ds.read_bag(p)
    frames = _frames(2)
    msgs = [(ds.IMAGE_TOPIC, im.stamp, im) for im, _, _ in frames]
    ds.write_bag(p, msgs)
    buf = p.read_bytes()
    p.write_bytes(buf[:len(buf) - 7])
    with pytest.raises(ds.BagFormatError):
        ds.read_bag(p)


def test_messages_are_replayed_in_receipt_time_order(tmp_path):
    """rosbag::View iterates in time order across chunks, whatever the order the records were
    written in; read_bag(time_order=False) exposes the file order."""
    from dbot_ros_amd import dataset as ds
    ims = []
    for k in (2, 0, 1):                     # written out of order
        im = ds.Image(np.full((4, 6), 0.5 + k, np.float32), ds.Stamp(10 + k, 0))
        ims.append(im)
    path = tmp_path / "o.bag"
    ds.write_bag(str(path), [(ds.IMAGE_TOPIC, im.stamp, im) for im in ims])
    in_file = [m[2][0] for m in ds.read_bag(str(path), time_order=False)]
    in_time = [m[2][0] for m in ds.read_bag(str(path))]
    assert in_file == [12, 10, 11] and in_time == [10, 11, 12]


def test_hand_assembled_bag_fixture_is_read_without_this_modules_writer():
    """tests/golden/dataset_fixture/ was laid down byte by byte from the published bag v2.0 description by
    tests/golden/make_bag_fixture.py, which does not import dbot_ros_amd.dataset (a reader bug mirrored
    in the module's own writer is invisible to a round trip).  One chunk, one connection per topic -- the reference's topic
    and file names, R:source/dbot_ros/util/tracking_dataset.cpp:92-99 --, two frames, index and chunk-info records, the
    4 096-byte bag header; ground_truth.txt in StoreTextFile's text (:326-360)."""
    import os
    import struct
    from dbot_ros_amd import dataset as ds
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_fixture")
    raw = open(os.path.join(here, "measurements.bag"), "rb").read()
    assert raw.startswith(b"#ROSBAG V2.0\n") and len(raw) == 6616
    # the fixture really has the structure it claims: the bag header's index_pos points at the first record behind the chunk's index
    hlen = struct.unpack_from("<I", raw, 13)[0]
    fields = ds._parse_header(raw[17:17 + hlen])
    index_pos = struct.unpack("<Q", fields["index_pos"])[0]
    assert struct.unpack("<I", fields["conn_count"])[0] == 2 and struct.unpack("<I", fields["chunk_count"])[0] == 1
    tail = list(ds._records(raw, index_pos))
    assert [h["op"][0] for h, _ in tail] == [7, 7, 6]
    d = ds.TrackingDataset(here)
    assert d.size() == 2
    want0 = np.array([0.50, 0.75, 1.00, 1.25, 1.50, np.nan, 2.00, 2.25, 2.50, 2.75, 3.00, 3.25], np.float32)
    want1 = np.array([0.625, 0.875, np.nan, 1.375, 1.625, 1.875, 2.125, 2.375, 2.625, 2.875, 3.125, 3.375], np.float32)
    for k, want in enumerate((want0, want1)):
        v = d.frame_vector(k)
        assert v.dtype == np.float32 and v.shape == (12,)
        assert np.array_equal(np.isnan(v), np.isnan(want)) and np.array_equal(v[~np.isnan(v)], want[~np.isnan(want)])
        assert (d.get_image(k).height, d.get_image(k).width) == (3, 4) and d.get_image(k).seq == k
        assert d.get_info(k).distortion_model == "plumb_bob" and len(d.get_info(k).D) == 5
    assert d.get_image(0).stamp == ds.Stamp(1400000000, 250000000) and str(d.get_image(1).stamp) == "1400000000.283333333"
    assert np.array_equal(d.get_camera_matrix(), np.array([[570.25, 0, 1.5], [0, 571.5, 1.0], [0, 0, 1.0]]))
    assert np.allclose(d.get_ground_truth(0), [0.01, -0.02, 0.70, 0.3, -0.5, 0.2, 0, 0, 0, 0, 0, 0])
    assert np.allclose(d.get_ground_truth(1), [0.012, -0.02, 0.70, 0.3, -0.4825, 0.2, 0.06, 0, 0, 0, 0.5236, 0])
    # the reference's own LoadTextFile reads ONE line: frame 1 then has no ground truth
    d1 = ds.TrackingDataset(here, load=False)
    d1.load(first_line_only=True)
    assert d1.get_ground_truth(0).size == 12 and d1.get_ground_truth(1).size == 0
    # file order is message order here; reading by time gives the same sequence
    msgs = ds.read_bag(os.path.join(here, "measurements.bag"))
    assert [m[0] for m in msgs] == ["XTION/depth/image", "XTION/depth/camera_info"] * 2
