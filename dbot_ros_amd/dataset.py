"""ROS-free reader/writer of dbot_ros's tracking-dataset format.

A dataset is a directory holding

  measurements.bag    ROS bag v2.0; sensor_msgs/Image on ``XTION/depth/image`` (32FC1 depth in
                      metres, NaN = no reading) and sensor_msgs/CameraInfo on
                      ``XTION/depth/camera_info`` (either topic also with a leading '/')
  ground_truth.txt    lines ``stamp s0 s1 ...`` (seconds, then the state vector)

which is what ``TrackingDataset::Load / Store`` read and write through the rosbag API
(R:source/dbot_ros/util/tracking_dataset.cpp:92-99 names and topics, :176-229 Load, :231-283 text
file, :326-360 Store).  Neither rosbag nor any ROS package exists in the build image, so this module
restates the published bag v2.0 container (records = <header_len><header><data_len><data>,
header = <field_len>name=value ..., op 0x03 bag header padded to 4 096 bytes, 0x05 chunk,
0x07 connection, 0x02 message data, 0x04 index data, 0x06 chunk info) and the ROS1 wire
serialisation of the two message types; there is no upstream bag in the reference tree to pin
it against (the repo ships none).  The tests pin the reader two ways: the round trip through this
module's own writer, and tests/golden/dataset_fixture/ -- a bag assembled byte by byte from the
published format by a script that does not import this module (tests/golden/make_bag_fixture.py).

Host-side data plumbing only: frames come out as float32 [rows*cols] in the layout
``RbSensor.set_observation`` takes (row-major, metres, NaN = no reading), K as the 3x3 matrix
``CameraData`` takes.
"""
import bz2
import os
import struct

import numpy as np

IMAGE_TOPIC = "XTION/depth/image"            # R:source/dbot_ros/util/tracking_dataset.cpp:94
INFO_TOPIC = "XTION/depth/camera_info"       # :95
OBSERVATIONS_FILENAME = "measurements.bag"   # :96
GROUND_TRUTH_FILENAME = "ground_truth.txt"   # :97
ADMISSIBLE_DELTA_TIME = 0.02                 # :98

_MAGIC = b"#ROSBAG V2.0\n"
_OP_MSG, _OP_BAG_HEADER, _OP_INDEX, _OP_CHUNK, _OP_CHUNK_INFO, _OP_CONNECTION = 2, 3, 4, 5, 6, 7
_IMAGE_TYPE, _IMAGE_MD5 = "sensor_msgs/Image", "060021388200f6f0f447d0fcd9c64743"
_INFO_TYPE, _INFO_MD5 = "sensor_msgs/CameraInfo", "c9a58c1b0b154e0e6da7578cb991d214"


class BagFormatError(ValueError):
    pass


# ------------------------------------------------------------------ record layer
def _parse_header(buf):
    fields, pos = {}, 0
    while pos < len(buf):
        if pos + 4 > len(buf):
            raise BagFormatError("truncated record header")
        (flen,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        field = buf[pos:pos + flen]
        if len(field) != flen or b"=" not in field:
            raise BagFormatError("malformed header field")
        name, value = field.split(b"=", 1)
        fields[name.decode()] = value
        pos += flen
    return fields


def _records(buf, pos=0):
    """Yield (header fields, data) of the records laid end to end in ``buf``."""
    n = len(buf)
    while pos < n:
        if pos + 4 > n:
            raise BagFormatError("truncated record")
        (hlen,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        header = _parse_header(buf[pos:pos + hlen])
        pos += hlen
        if pos + 4 > n:
            raise BagFormatError("truncated record")
        (dlen,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        data = buf[pos:pos + dlen]
        if len(data) != dlen:
            raise BagFormatError("truncated record data")
        pos += dlen
        yield header, data


def _pack_header(fields):
    out = b""
    for name, value in fields:
        field = name.encode() + b"=" + value
        out += struct.pack("<I", len(field)) + field
    return out


def _pack_record(fields, data):
    h = _pack_header(fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


# ------------------------------------------------------------------ message layer
class Stamp(tuple):
    """(secs, nsecs); prints and converts like ros::Time."""

    def __new__(cls, secs, nsecs=0):
        return super().__new__(cls, (int(secs), int(nsecs)))

    @classmethod
    def from_sec(cls, t):
        secs = int(np.floor(t))
        nsecs = int(round((t - secs) * 1e9))
        if nsecs >= 1000000000:
            secs, nsecs = secs + 1, nsecs - 1000000000
        return cls(secs, nsecs)

    def to_sec(self):
        return self[0] + 1e-9 * self[1]

    def __str__(self):                                # ros::Time's operator<<
        return "%d.%09d" % (self[0], self[1])


def _rd_string(buf, pos):
    (n,) = struct.unpack_from("<I", buf, pos)
    return buf[pos + 4:pos + 4 + n].decode(), pos + 4 + n


def _rd_header(buf, pos):
    seq, secs, nsecs = struct.unpack_from("<III", buf, pos)
    frame_id, pos = _rd_string(buf, pos + 12)
    return {"seq": seq, "stamp": Stamp(secs, nsecs), "frame_id": frame_id}, pos


def _wr_string(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def _wr_header(seq, stamp, frame_id):
    return struct.pack("<III", seq, stamp[0], stamp[1]) + _wr_string(frame_id)


class Image:
    """sensor_msgs/Image restricted to what the tracker consumes: one float channel."""

    def __init__(self, depth, stamp, frame_id="XTION", seq=0):
        depth = np.asarray(depth)
        if depth.ndim != 2:
            raise ValueError("depth image must be [rows, cols]")
        self.depth = np.ascontiguousarray(depth, dtype=np.float32)
        self.stamp = stamp if isinstance(stamp, Stamp) else Stamp.from_sec(stamp)
        self.frame_id, self.seq = frame_id, seq

    height = property(lambda self: self.depth.shape[0])
    width = property(lambda self: self.depth.shape[1])

    def serialize(self):
        data = self.depth.astype("<f4").tobytes()
        return (_wr_header(self.seq, self.stamp, self.frame_id) + struct.pack("<II", self.height, self.width) +
                _wr_string("32FC1") + struct.pack("<BI", 0, 4 * self.width) + struct.pack("<I", len(data)) + data)

    @classmethod
    def deserialize(cls, buf):
        hdr, pos = _rd_header(buf, 0)
        height, width = struct.unpack_from("<II", buf, pos)
        encoding, pos = _rd_string(buf, pos + 8)
        bigendian, step = struct.unpack_from("<BI", buf, pos)
        (n,) = struct.unpack_from("<I", buf, pos + 5)
        raw = buf[pos + 9:pos + 9 + n]
        if len(raw) != n or n < step * height:
            raise BagFormatError("truncated image data")
        if encoding == "32FC1":                       # cv_image.at<float>: metres
            px = np.frombuffer(raw, dtype=">f4" if bigendian else "<f4", count=(step // 4) * height)
            depth = px.reshape(height, step // 4)[:, :width].astype(np.float32)
        elif encoding in ("16UC1", "mono16"):         # OpenNI raw depth: millimetres, 0 = no reading
            px = np.frombuffer(raw, dtype=">u2" if bigendian else "<u2", count=(step // 2) * height)
            mm = px.reshape(height, step // 2)[:, :width]
            depth = np.where(mm == 0, np.nan, mm.astype(np.float32) * np.float32(1e-3)).astype(np.float32)
        else:
            raise BagFormatError(f"unsupported depth encoding {encoding!r}")
        return cls(depth, hdr["stamp"], hdr["frame_id"], hdr["seq"])

    def to_vector(self, downsampling=1):
        """ri::to_eigen_vector: eval(row, col) = native(row*f, col*f), row-major
        (R:source/dbot_ros/util/ros_interface.h:152-168)."""
        f = int(downsampling)
        rows, cols = self.height // f, self.width // f
        return np.ascontiguousarray(self.depth[:rows * f:f, :cols * f:f]).reshape(rows * cols)


class CameraInfo:
    def __init__(self, K, height, width, stamp, frame_id="XTION", seq=0, distortion_model="plumb_bob", D=(),
                 R=None, P=None):
        self.K = np.asarray(K, dtype=np.float64).reshape(3, 3)
        self.height, self.width = int(height), int(width)
        self.stamp = stamp if isinstance(stamp, Stamp) else Stamp.from_sec(stamp)
        self.frame_id, self.seq, self.distortion_model = frame_id, seq, distortion_model
        self.D = np.asarray(D, dtype=np.float64).ravel()
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3)
        self.P = np.hstack([self.K, np.zeros((3, 1))]) if P is None else np.asarray(P, dtype=np.float64).reshape(3, 4)

    def serialize(self):
        return (_wr_header(self.seq, self.stamp, self.frame_id) + struct.pack("<II", self.height, self.width) +
                _wr_string(self.distortion_model) + struct.pack("<I", len(self.D)) + self.D.astype("<f8").tobytes() +
                self.K.astype("<f8").tobytes() + self.R.astype("<f8").tobytes() + self.P.astype("<f8").tobytes() +
                struct.pack("<II", 0, 0) + struct.pack("<IIIIB", 0, 0, 0, 0, 0))

    @classmethod
    def deserialize(cls, buf):
        hdr, pos = _rd_header(buf, 0)
        height, width = struct.unpack_from("<II", buf, pos)
        model, pos = _rd_string(buf, pos + 8)
        (nd,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        D = np.frombuffer(buf, dtype="<f8", count=nd, offset=pos)
        pos += 8 * nd
        if pos + 8 * 30 > len(buf):
            raise BagFormatError("truncated camera info")
        K = np.frombuffer(buf, dtype="<f8", count=9, offset=pos)
        R = np.frombuffer(buf, dtype="<f8", count=9, offset=pos + 72)
        P = np.frombuffer(buf, dtype="<f8", count=12, offset=pos + 144)
        return cls(K, height, width, hdr["stamp"], hdr["frame_id"], hdr["seq"], model, D, R, P)


_TYPES = {_IMAGE_TYPE: Image, _INFO_TYPE: CameraInfo}


# ------------------------------------------------------------------ bag layer
def read_bag(path, topics=None, time_order=True):
    """All messages of a v2.0 bag as (topic, type, receipt Stamp, raw bytes), in RECEIPT-TIME order
    (stable: file order among equal stamps), which is how rosbag::View iterates across chunks
    (R:source/dbot_ros/util/tracking_dataset.cpp:189,204); time_order=False gives file order.
    Chunks may be uncompressed or bz2; the index records are not needed and are skipped."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if not buf.startswith(_MAGIC):
        raise BagFormatError("not a ROS bag v2.0 file")
    connections, out = {}, []

    def handle(header, data):
        op = header.get("op", b"\xff")[0]
        if op == _OP_CONNECTION:
            (conn,) = struct.unpack("<I", header["conn"])
            info = _parse_header(data)
            connections[conn] = (header["topic"].decode(), info.get("type", b"").decode())
        elif op == _OP_MSG:
            (conn,) = struct.unpack("<I", header["conn"])
            secs, nsecs = struct.unpack("<II", header["time"])
            if conn not in connections:
                raise BagFormatError(f"message on unknown connection {conn}")
            topic, mtype = connections[conn]
            if topics is None or topic in topics:
                out.append((topic, mtype, Stamp(secs, nsecs), data))

    for header, data in _records(buf, len(_MAGIC)):
        op = header.get("op", b"\xff")[0]
        if op == _OP_CHUNK:
            comp = header.get("compression", b"none").decode()
            if comp == "bz2":
                data = bz2.decompress(data)
            elif comp != "none":
                raise BagFormatError(f"unsupported chunk compression {comp!r}")
            (size,) = struct.unpack("<I", header["size"])
            if len(data) != size:
                raise BagFormatError("chunk size mismatch")
            for h2, d2 in _records(data):
                handle(h2, d2)
        else:
            handle(header, data)
    if time_order:
        out.sort(key=lambda m: (m[2][0], m[2][1]))     # (secs, nsecs); list.sort is stable
    return out


def write_bag(path, messages, compression="none", chunk_messages=16):
    """messages: iterable of (topic, receipt Stamp, Image | CameraInfo).  Writes a complete v2.0
    bag (bag header, chunks with their index records, connection and chunk-info records)."""
    conns, order = {}, []
    for topic, stamp, msg in messages:
        mtype, md5 = (_IMAGE_TYPE, _IMAGE_MD5) if isinstance(msg, Image) else (_INFO_TYPE, _INFO_MD5)
        if topic not in conns:
            conns[topic] = (len(conns), mtype, md5)
        order.append((conns[topic][0], stamp, msg.serialize()))

    def conn_record(topic):
        cid, mtype, md5 = conns[topic]
        info = _pack_header([("topic", topic.encode()), ("type", mtype.encode()), ("md5sum", md5.encode()),
                             ("message_definition", b"")])
        return _pack_record([("op", bytes([_OP_CONNECTION])), ("conn", struct.pack("<I", cid)),
                             ("topic", topic.encode())], info)

    by_id = {cid: topic for topic, (cid, _, _) in conns.items()}
    body, chunk_infos, seen = b"", [], set()
    base = len(_MAGIC) + 4096
    for c0 in range(0, len(order), chunk_messages):
        part = order[c0:c0 + chunk_messages]
        raw, index = b"", {}
        for cid, stamp, payload in part:
            if cid not in seen:
                seen.add(cid)
                raw += conn_record(by_id[cid])
            index.setdefault(cid, []).append((stamp, len(raw)))
            raw += _pack_record([("op", bytes([_OP_MSG])), ("conn", struct.pack("<I", cid)),
                                 ("time", struct.pack("<II", *stamp))], payload)
        data = bz2.compress(raw) if compression == "bz2" else raw
        chunk_pos = base + len(body)
        body += _pack_record([("op", bytes([_OP_CHUNK])), ("compression", compression.encode()),
                              ("size", struct.pack("<I", len(raw)))], data)
        for cid, entries in index.items():
            idx = b"".join(struct.pack("<III", s[0], s[1], off) for s, off in entries)
            body += _pack_record([("op", bytes([_OP_INDEX])), ("ver", struct.pack("<I", 1)),
                                  ("conn", struct.pack("<I", cid)), ("count", struct.pack("<I", len(entries)))], idx)
        stamps = [s for _, s, _ in part]
        counts = b"".join(struct.pack("<II", cid, len(e)) for cid, e in index.items())
        chunk_infos.append(_pack_record([("op", bytes([_OP_CHUNK_INFO])), ("ver", struct.pack("<I", 1)),
                                         ("chunk_pos", struct.pack("<Q", chunk_pos)),
                                         ("start_time", struct.pack("<II", *min(stamps))),
                                         ("end_time", struct.pack("<II", *max(stamps))),
                                         ("count", struct.pack("<I", len(index)))], counts))
    index_pos = base + len(body)
    tail = b"".join(conn_record(t) for t in conns) + b"".join(chunk_infos)
    hdr = _pack_header([("op", bytes([_OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", index_pos)),
                        ("conn_count", struct.pack("<I", len(conns))),
                        ("chunk_count", struct.pack("<I", len(chunk_infos)))])
    pad = 4096 - 4 - len(hdr) - 4
    with open(path, "wb") as fh:
        fh.write(_MAGIC + struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", pad) + b" " * pad + body + tail)


# ------------------------------------------------------------------ dataset
class DataFrame:
    def __init__(self, image, info):
        self.image, self.info = image, info
        self.ground_truth = np.zeros(0)
        self.deviation = np.zeros(0)


class TrackingDataset:
    """TrackingDataset of R:source/dbot_ros/util/tracking_dataset.cpp: same file names, topics,
    accessors (snake_case) and admissible stamp difference."""

    def __init__(self, path, load=True):
        self.path = str(path)
        self.image_topic, self.info_topic = IMAGE_TOPIC, INFO_TOPIC
        self.observations_filename, self.ground_truth_filename = OBSERVATIONS_FILENAME, GROUND_TRUTH_FILENAME
        self.admissible_delta_time = ADMISSIBLE_DELTA_TIME
        self.data = []
        if load:
            self.load()

    # -- TimeSynchronizer<Image, CameraInfo>(queue 25): exact header-stamp matches, in order
    def _synchronize(self, messages, queue=25):
        images, infos = [], []
        for topic, mtype, _, raw in messages:
            t = topic.lstrip("/")
            if t == self.image_topic and mtype == _IMAGE_TYPE:
                images.append(Image.deserialize(raw))
            elif t == self.info_topic and mtype == _INFO_TYPE:
                infos.append(CameraInfo.deserialize(raw))
            else:
                continue
            images, infos = images[-queue:], infos[-queue:]
            stamps = {m.stamp for m in infos}
            for im in [m for m in images if m.stamp in stamps]:
                info = next(m for m in infos if m.stamp == im.stamp)
                self.add_frame(im, info)
                images = [m for m in images if m.stamp > im.stamp]
                infos = [m for m in infos if m.stamp > im.stamp]

    def add_frame(self, image, info, ground_truth=None, deviation=None):
        f = DataFrame(image, info)
        if ground_truth is not None:
            f.ground_truth = np.asarray(ground_truth, dtype=np.float64).ravel()
        if deviation is not None:
            f.deviation = np.asarray(deviation, dtype=np.float64).ravel()
        self.data.append(f)

    def load(self, first_line_only=False):
        """Messages are replayed in receipt-time order, as rosbag::View delivers them.
        ground_truth.txt: the reference's LoadTextFile (:231-283) does ONE getline, i.e. attaches
        only the first line's state (to the frames within admissible_delta_time of its stamp);
        first_line_only=True reproduces that as written.  The default reads every line -- a
        deliberate divergence: Store() (:298-333) writes one line per frame, and a replay that
        wants per-frame ground truth needs them all."""
        topics = {self.image_topic, self.info_topic, "/" + self.image_topic, "/" + self.info_topic}
        self._synchronize(read_bag(os.path.join(self.path, self.observations_filename), topics))
        gt = os.path.join(self.path, self.ground_truth_filename)
        if os.path.exists(gt):
            self.load_text_file(gt, "ground_truth", first_line_only)

    def load_text_file(self, filename, kind="ground_truth", first_line_only=False):
        with open(filename) as fh:
            lines = [ln for ln in fh.read().splitlines() if ln.strip()]
        for ln in lines[:1] if first_line_only else lines:
            vals = [float(x) for x in ln.split()]
            stamp, state = vals[0], np.array(vals[1:], dtype=np.float64)
            for f in self.data:
                if abs(f.image.stamp.to_sec() - stamp) <= self.admissible_delta_time:
                    setattr(f, kind, state)

    def store(self):
        """Store(): refuses to overwrite (:289-296)."""
        bag = os.path.join(self.path, self.observations_filename)
        gt = os.path.join(self.path, self.ground_truth_filename)
        if os.path.exists(bag) or os.path.exists(gt):
            raise FileExistsError(f"TrackingDataset {self.path} already exists, will not overwrite")
        os.makedirs(self.path, exist_ok=True)
        msgs = []
        for f in self.data:
            msgs.append((self.image_topic, f.image.stamp, f.image))
            msgs.append((self.info_topic, f.info.stamp, f.info))
        write_bag(bag, msgs)
        with open(gt, "w") as fh:
            for f in self.data:
                if f.ground_truth.size:
                    fh.write(str(f.image.stamp) + " " + " ".join(repr(float(x)) for x in f.ground_truth) + "\n")

    # -- accessors
    def size(self):
        return len(self.data)

    __len__ = size

    def get_image(self, index):
        return self.data[index].image

    def get_info(self, index):
        return self.data[index].info

    def get_camera_matrix(self, index=0):
        """GetCameraMatrix: always frame 0's K, whatever the index (:157-164)."""
        return self.data[0].info.K.copy()

    def get_ground_truth(self, index):
        return self.data[index].ground_truth

    def frame_vector(self, index, downsampling=1):
        """The frame as RbSensor.set_observation takes it (float32, row-major, metres)."""
        return self.data[index].image.to_vector(downsampling)
