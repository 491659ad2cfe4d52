"""Hand-assembles tests/golden/dataset_fixture/{measurements.bag, ground_truth.txt} byte by byte from the PUBLISHED
description of the ROS bag format 2.0 (wiki.ros.org/Bags/Format/2.0) and of the ROS1 wire serialisation of
sensor_msgs/Image and sensor_msgs/CameraInfo (their .msg definitions) -- WITHOUT importing dbot_ros_amd.dataset: a
reader bug mirrored in that module's own writer is invisible to a writer -> reader round trip, it is
not invisible to bytes laid down field by field here.

Layout (what `rosbag record` produces for two depth frames, uncompressed):
  "#ROSBAG V2.0\\n"
  bag header record   op=0x03  index_pos conn_count=2 chunk_count=1, padded with spaces to 4096 bytes
  chunk record        op=0x05  compression=none size=<bytes>; its data = connection 0, connection 1, then per frame an
                               Image message record and a CameraInfo message record (op=0x02 conn time)
  index data records  op=0x04  ver=1 conn count, one per connection: (time, offset in the chunk) per message
  -- index_pos points here --
  connection records  op=0x07  once more, outside the chunk
  chunk info record   op=0x06  ver=1 chunk_pos start_time end_time count=2: (conn, message count) per connection
Header fields are written in name order, as rosbag's std::map does.  Topic and file names are the reference's
(R:source/dbot_ros/util/tracking_dataset.cpp:92-99); ground_truth.txt is the text StoreTextFile writes (:326-360):
`<ros::Time stamp: sec.nnnnnnnnn> <state vector, space separated>` per frame.

Run:  python tests/golden/make_bag_fixture.py    (rewrites the two files; they are committed)
"""
import os
import struct

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset_fixture")

ROWS, COLS = 3, 4
NAN = float("nan")
# two 3x4 depth frames, metres; NaN = no reading.  Frame 1 differs from frame 0 in every finite pixel.
FRAMES = [
    [0.50, 0.75, 1.00, 1.25, 1.50, NAN, 2.00, 2.25, 2.50, 2.75, 3.00, 3.25],
    [0.625, 0.875, NAN, 1.375, 1.625, 1.875, 2.125, 2.375, 2.625, 2.875, 3.125, 3.375],
]
STAMPS = [(1400000000, 250000000), (1400000000, 283333333)]        # (sec, nsec): 30 Hz apart
RECEIPT = [(1400000000, 251000000), (1400000000, 284333333)]       # receipt times of the records (1 ms later)
K = [570.25, 0.0, 1.5, 0.0, 571.5, 1.0, 0.0, 0.0, 1.0]
GROUND_TRUTH = [
    [0.01, -0.02, 0.70, 0.3, -0.5, 0.2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
    [0.012, -0.02, 0.70, 0.3, -0.4825, 0.2, 0.06, 0.0, 0.0, 0.0, 0.5236, 0.0],
]

IMAGE_TOPIC, INFO_TOPIC = "XTION/depth/image", "XTION/depth/camera_info"
IMAGE_DEF = b"# sensor_msgs/Image (definition text is not interpreted by readers)\n"
INFO_DEF = b"# sensor_msgs/CameraInfo (definition text is not interpreted by readers)\n"


def u32(v):
    return struct.pack("<I", v)


def string(b):
    return u32(len(b)) + b


def header(fields):
    """<field_len><name>=<value> for every field, names in sorted order."""
    out = b""
    for name in sorted(fields):
        f = name.encode() + b"=" + fields[name]
        out += u32(len(f)) + f
    return out


def record(fields, data):
    h = header(fields)
    return u32(len(h)) + h + u32(len(data)) + data


def ros_time(t):
    return struct.pack("<II", t[0], t[1])


def std_header(seq, stamp, frame_id):
    return u32(seq) + ros_time(stamp) + string(frame_id)


def image_msg(seq, stamp, px):
    data = b"".join(struct.pack("<f", v) for v in px)
    return (std_header(seq, stamp, b"XTION") + u32(ROWS) + u32(COLS) + string(b"32FC1") + struct.pack("<B", 0) +
            u32(4 * COLS) + u32(len(data)) + data)


def info_msg(seq, stamp):
    f8 = lambda vals: b"".join(struct.pack("<d", v) for v in vals)      # noqa: E731
    D = [0.0, 0.0, 0.0, 0.0, 0.0]
    R = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
    P = [K[0], K[1], K[2], 0.0, K[3], K[4], K[5], 0.0, K[6], K[7], K[8], 0.0]
    roi = u32(0) + u32(0) + u32(0) + u32(0) + struct.pack("<B", 0)
    return (std_header(seq, stamp, b"XTION") + u32(ROWS) + u32(COLS) + string(b"plumb_bob") + u32(len(D)) + f8(D) +
            f8(K) + f8(R) + f8(P) + u32(1) + u32(1) + roi)


def connection(conn, topic, mtype, md5, definition):
    data = header({"topic": topic, "type": mtype, "md5sum": md5, "message_definition": definition, "callerid": b"/fixture"})
    return record({"op": b"\x07", "conn": u32(conn), "topic": topic}, data)


def main():
    conns = [connection(0, IMAGE_TOPIC.encode(), b"sensor_msgs/Image", b"060021388200f6f0f447d0fcd9c64743", IMAGE_DEF),
             connection(1, INFO_TOPIC.encode(), b"sensor_msgs/CameraInfo", b"c9a58c1b0b154e0e6da7578cb991d214", INFO_DEF)]
    chunk = b"".join(conns)
    offsets = {0: [], 1: []}
    for k in range(2):
        for conn, payload in ((0, image_msg(k, STAMPS[k], FRAMES[k])), (1, info_msg(k, STAMPS[k]))):
            offsets[conn].append((RECEIPT[k], len(chunk)))
            chunk += record({"op": b"\x02", "conn": u32(conn), "time": ros_time(RECEIPT[k])}, payload)
    chunk_pos = len(b"#ROSBAG V2.0\n") + 4096
    body = record({"op": b"\x05", "compression": b"none", "size": u32(len(chunk))}, chunk)
    for conn in (0, 1):
        idx = b"".join(ros_time(t) + u32(off) for t, off in offsets[conn])
        body += record({"op": b"\x04", "ver": u32(1), "conn": u32(conn), "count": u32(len(offsets[conn]))}, idx)
    index_pos = chunk_pos + len(body)
    tail = b"".join(conns)
    tail += record({"op": b"\x06", "ver": u32(1), "chunk_pos": struct.pack("<Q", chunk_pos), "start_time": ros_time(RECEIPT[0]),
                    "end_time": ros_time(RECEIPT[1]), "count": u32(2)}, u32(0) + u32(2) + u32(1) + u32(2))
    bag_header = header({"op": b"\x03", "index_pos": struct.pack("<Q", index_pos), "conn_count": u32(2), "chunk_count": u32(1)})
    pad = 4096 - 4 - len(bag_header) - 4          # the record (both length words included) occupies 4096 bytes
    first = u32(len(bag_header)) + bag_header + u32(pad) + b" " * pad
    assert len(first) == 4096
    os.makedirs(HERE, exist_ok=True)
    with open(os.path.join(HERE, "measurements.bag"), "wb") as f:
        f.write(b"#ROSBAG V2.0\n" + first + body + tail)
    with open(os.path.join(HERE, "ground_truth.txt"), "w") as f:
        for st, gt in zip(STAMPS, GROUND_TRUTH):
            # ros::Time's operator<< : sec, '.', nsec zero-padded to 9; Eigen's default stream format for the transposed vector
            f.write("%d.%09d " % st + " ".join("%g" % v for v in gt) + "\n")


if __name__ == "__main__":
    main()
