"""rosbag-v2 reader / writer (cerberus_amd/host/vilo_rosbag.*) and the message-level replay that stands in for the
reference's node (src/main.cpp:204-330, 415-437): the byte layout against the published format, every message type through a write /
read round trip bit for bit, the node's pairing and contact selection, and a synthetic sensor stream replayed from a bag against the
same messages fed from memory — identical intervals on the CPU, identical trajectory on the GPU."""
import struct

import numpy as np
import pytest


def _frames(cfg, n, seed=5):
    from cerberus_amd import sequence
    stream = sequence.Stream(cfg, seed=seed)
    return stream, [stream.next() for _ in range(n)]


def test_byte_layout_of_a_minimal_bag(tmp_path):
    """One IMU message: magic, the 4096-byte bag header record, one uncompressed chunk holding a connection record and a message record,
    the chunk's index record, then — at index_pos — the connection record again and the chunk info (Bags/Format/2.0)."""
    from cerberus_amd import rosbag as rb
    p = tmp_path / "one.bag"
    with rb.BagWriter(p) as w:
        w.write(dict(kind=rb.KIND_IMU, topic="/imu", seq=7, secs=12, nsecs=345, frame_id="b", linear_acceleration=[1.0, 2.0, 3.0], angular_velocity=[4.0, 5.0, 6.0]))
    raw = p.read_bytes()
    assert raw[:13] == b"#ROSBAG V2.0\n"

    def record(at):
        hl = struct.unpack_from("<I", raw, at)[0]
        hdr = raw[at + 4:at + 4 + hl]
        dl = struct.unpack_from("<I", raw, at + 4 + hl)[0]
        data = raw[at + 8 + hl:at + 8 + hl + dl]
        fields, q = {}, 0
        while q < hl:
            fl = struct.unpack_from("<I", hdr, q)[0]
            name, val = hdr[q + 4:q + 4 + fl].split(b"=", 1)
            fields[name.decode()] = val
            q += 4 + fl
        return fields, data, at + 8 + hl + dl

    f, d, at = record(13)
    assert f["op"] == b"\x03" and at == 13 + 4096 and set(d) == {0x20}                               # bag header, space-padded to 4096 bytes
    index_pos = struct.unpack("<Q", f["index_pos"])[0]
    assert struct.unpack("<I", f["conn_count"])[0] == 1 and struct.unpack("<I", f["chunk_count"])[0] == 1
    f, chunk, at = record(at)
    assert f["op"] == b"\x05" and f["compression"] == b"none" and struct.unpack("<I", f["size"])[0] == len(chunk)
    chunk_pos = 13 + 4096
    # inside the chunk: connection record, then the message
    hl = struct.unpack_from("<I", chunk, 0)[0]
    assert b"op=\x07" in chunk[4:4 + hl] and b"topic=/imu" in chunk[4:4 + hl]
    dl = struct.unpack_from("<I", chunk, 4 + hl)[0]
    conn_data = chunk[8 + hl:8 + hl + dl]
    assert b"type=sensor_msgs/Imu" in conn_data and b"md5sum=6a62c6daae103f4ff57a132d6f95cec2" in conn_data and b"message_definition=" in conn_data
    msg_at = 8 + hl + dl
    hl2 = struct.unpack_from("<I", chunk, msg_at)[0]
    mh = chunk[msg_at + 4:msg_at + 4 + hl2]
    assert b"op=\x02" in mh and b"time=" + struct.pack("<II", 12, 345) in mh and b"conn=" + struct.pack("<I", 0) in mh
    body = chunk[msg_at + 8 + hl2:]
    # sensor_msgs/Imu in ROS 1 serialisation: Header (seq, stamp, frame_id), quaternion, cov[9], angular velocity, cov[9], acceleration, cov[9]
    assert body[:12] == struct.pack("<III", 7, 12, 345) and body[12:17] == struct.pack("<I", 1) + b"b"
    vals = struct.unpack("<37d", body[17:])
    assert vals[0:4] == (0.0, 0.0, 0.0, 1.0) and vals[13:16] == (4.0, 5.0, 6.0) and vals[25:28] == (1.0, 2.0, 3.0) and len(body) == 17 + 37 * 8
    f, d, at = record(at)
    assert f["op"] == b"\x04" and struct.unpack("<I", f["ver"])[0] == 1 and struct.unpack("<I", f["count"])[0] == 1          # index data of the chunk
    assert struct.unpack("<IIi", d) == (12, 345, msg_at)                                                                      # time, offset inside the chunk
    assert at == index_pos
    f, d, at = record(at)
    assert f["op"] == b"\x07"
    f, d, at = record(at)
    assert f["op"] == b"\x06" and struct.unpack("<Q", f["chunk_pos"])[0] == chunk_pos and struct.unpack("<II", d) == (0, 1)
    assert struct.unpack("<II", f["start_time"]) == (12, 345) and struct.unpack("<II", f["end_time"]) == (12, 345)
    assert at == len(raw)


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_every_message_type_round_trips_bit_for_bit(tmp_path, compression):
    from cerberus_amd import rosbag as rb
    rng = np.random.default_rng(3)
    msgs = []
    for i in range(300):
        secs, nsecs = 100 + i // 7, (i * 142857143) % 1000000000
        k = i % 4
        if k == 0:
            msgs.append(dict(kind=rb.KIND_IMU, topic="/a/imu", seq=i, secs=secs, nsecs=nsecs, frame_id="imu_link", linear_acceleration=rng.normal(size=3),
                             angular_velocity=rng.normal(size=3)))
        elif k == 1:
            n = 16 if i % 8 == 1 else 3
            msgs.append(dict(kind=rb.KIND_JOINT_STATE, topic="/a/joint_foot", seq=i, secs=secs, nsecs=nsecs, position=rng.normal(size=n), velocity=rng.normal(size=n),
                             effort=rng.normal(size=n)))
        elif k == 2:
            h, w = 5 + i % 3, 9
            msgs.append(dict(kind=rb.KIND_IMAGE, topic="/cam%d" % ((i // 4) % 2), seq=i, secs=secs, nsecs=nsecs, frame_id="cam", height=h, width=w, step=w + 3,
                             encoding="mono8", data=rng.integers(0, 256, size=h * (w + 3), dtype=np.uint8)))
        else:
            n = i % 11   # (an empty cloud too)
            msgs.append(dict(kind=rb.KIND_POINT_CLOUD, topic="/feature_tracker/feature", seq=i, secs=secs, nsecs=nsecs, points=rng.normal(size=(n, 3)).astype(np.float32),
                             channels=rng.normal(size=(6, n)).astype(np.float32), channel_names=["id", "camera_id", "p_u", "p_v", "velocity_x", "velocity_y"]))
    p = tmp_path / "all.bag"
    with rb.BagWriter(p, chunk_threshold=4096, compression=compression) as w:   # small chunks: many of them; a connection record only in the chunk of the connection's first message
        for m in msgs:
            w.write(m)
    r = rb.BagReader(p)
    assert r.conn_count == 5 and r.chunk_count > 5 and r.index_pos > 4096
    if compression == "none":
        assert p.read_bytes()[:r.index_pos].count(b"type=sensor_msgs/") == 5      # (the connection records inside the chunks: one per topic)
    assert p.read_bytes().count(b"compression=" + compression.encode()) == r.chunk_count
    back = list(r)
    assert len(back) == len(msgs)
    for a, b in zip(msgs, back):
        assert (a["kind"], a["topic"], a["seq"], a["secs"], a["nsecs"]) == (b["kind"], b["topic"], b["seq"], b["secs"], b["nsecs"])
        assert (b["rec_secs"], b["rec_nsecs"]) == (a["secs"], a["nsecs"])
        for key in ("linear_acceleration", "angular_velocity", "position", "velocity", "effort", "data", "points", "channels"):
            if key in a:
                np.testing.assert_array_equal(np.asarray(a[key]), b[key])
        for key in ("frame_id", "height", "width", "step", "encoding", "channel_names"):
            if key in a:
                assert a[key] == b[key], key
        assert b["type"] == {0: "sensor_msgs/Imu", 1: "sensor_msgs/JointState", 2: "sensor_msgs/Image", 3: "sensor_msgs/PointCloud"}[a["kind"]]


def test_damaged_compressed_and_foreign_bags_are_refused(tmp_path):
    from cerberus_amd import rosbag as rb
    p = tmp_path / "ok.bag"
    with rb.BagWriter(p) as w:
        for i in range(10):
            w.write(dict(kind=rb.KIND_IMU, topic="/imu", seq=i, secs=1, nsecs=i, linear_acceleration=np.zeros(3), angular_velocity=np.zeros(3)))
    raw = p.read_bytes()
    (tmp_path / "text.bag").write_bytes(b"not a bag at all")
    with pytest.raises(rb.BagError):
        rb.BagReader(tmp_path / "text.bag")
    with pytest.raises(rb.BagError):
        rb.BagReader(tmp_path / "missing.bag")
    (tmp_path / "cut.bag").write_bytes(raw[:13 + 4096 + 200])        # ends inside the chunk
    with pytest.raises(rb.BagError):
        list(rb.BagReader(tmp_path / "cut.bag"))
    assert raw.count(b"compression=none") == 1
    at = 13 + 4096                                                   # the chunk record: its header says lz4 now (and is one byte shorter)
    hl = struct.unpack_from("<I", raw, at)[0]
    hdr = raw[at + 4:at + 4 + hl].replace(b"\x10\x00\x00\x00compression=none", b"\x0f\x00\x00\x00compression=lz4")
    (tmp_path / "lz4.bag").write_bytes(raw[:at] + struct.pack("<I", len(hdr)) + hdr + raw[at + 4 + hl:])
    with pytest.raises(rb.BagError, match="damaged"):                # (the bytes are not an LZ4 frame)
        list(rb.BagReader(tmp_path / "lz4.bag"))
    hdr = raw[at + 4:at + 4 + hl].replace(b"\x10\x00\x00\x00compression=none", b"\x10\x00\x00\x00compression=zstd")
    (tmp_path / "zstd.bag").write_bytes(raw[:at] + struct.pack("<I", len(hdr)) + hdr + raw[at + 4 + hl:])
    with pytest.raises(rb.BagError, match="compress"):
        list(rb.BagReader(tmp_path / "zstd.bag"))
    # a topic of a type the reader does not know is passed over as KIND_OTHER, the rest of the bag still reads
    other = raw.replace(b"type=sensor_msgs/Imu", b"type=sensor_msgs/Gps")
    (tmp_path / "other.bag").write_bytes(other)
    kinds = [m["kind"] for m in rb.BagReader(tmp_path / "other.bag")]
    assert kinds == [rb.KIND_OTHER] * 10


def test_lengths_a_damaged_bag_claims_are_never_allocated(tmp_path):
    """Sizes in a bag come from the file. A record whose data length exceeds what the file still holds, and a compressed chunk whose header
    asks for gigabytes, are refused with the format error — before any allocation of that size, and as a return code through the C
    boundary (they used to end in std::bad_alloc / length_error inside ctypes)."""
    import resource
    from cerberus_amd import rosbag as rb
    p = tmp_path / "ok.bag"
    with rb.BagWriter(p, compression="bz2") as w:
        for i in range(10):
            w.write(dict(kind=rb.KIND_IMU, topic="/imu", seq=i, secs=1, nsecs=i, linear_acceleration=np.zeros(3), angular_velocity=np.zeros(3)))
    raw = p.read_bytes()
    at = 13 + 4096                                                   # the chunk record
    hl = struct.unpack_from("<I", raw, at)[0]
    # (1) the chunk's data length says 3 GiB
    huge = raw[:at + 4 + hl] + struct.pack("<I", 3 << 30) + raw[at + 8 + hl:]
    (tmp_path / "dl.bag").write_bytes(huge)
    # (2) the header's `size` field (uncompressed length of the bz2 stream) says 3.9 GiB
    hdr = raw[at + 4:at + 4 + hl]
    q = hdr.index(b"size=")
    hdr2 = hdr[:q + 5] + struct.pack("<I", 0xF0000000) + hdr[q + 9:]
    (tmp_path / "size.bag").write_bytes(raw[:at + 4] + hdr2 + raw[at + 4 + hl:])
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    for name in ("dl.bag", "size.bag"):
        with pytest.raises(rb.BagError, match="damaged|format"):
            list(rb.BagReader(tmp_path / name))
    assert resource.getrlimit(resource.RLIMIT_AS) == (soft, hard)


@pytest.mark.parametrize("compression", ["bz2", "lz4"])
def test_chunks_of_blank_images_read_back(tmp_path, compression):
    """A chunk of three black 640 x 480 mono8 images (921 672 bytes) is 160 bytes of bz2 — nearly 6000 : 1. The reader's guard against
    a lying `size` field must not take such a chunk for a lie (a bound of 1024 x the compressed bytes did): bz2 chunks are
    held to an absolute cap, LZ4 ones to the codec's own 255 : 1. Black, constant and saturated frames, default chunk threshold."""
    from cerberus_amd import rosbag as rb
    p = tmp_path / "black.bag"
    imgs = [np.zeros(640 * 480, np.uint8), np.zeros(640 * 480, np.uint8), np.zeros(640 * 480, np.uint8), np.full(640 * 480, 255, np.uint8),
            np.full(640 * 480, 17, np.uint8), np.zeros(640 * 480, np.uint8)]
    with rb.BagWriter(p, compression=compression) as w:
        for i, im in enumerate(imgs):
            w.write(dict(kind=rb.KIND_IMAGE, topic="/cam0", seq=i, secs=1, nsecs=i, frame_id="cam", height=480, width=640, step=640, encoding="mono8", data=im))
    assert p.stat().st_size < 40000          # (the point: thousands to one)
    back = [m for m in rb.BagReader(p) if m["kind"] == rb.KIND_IMAGE]
    assert len(back) == len(imgs)
    for m, im in zip(back, imgs):
        assert (m["height"], m["width"]) == (480, 640) and np.array_equal(np.frombuffer(bytes(m["data"]), np.uint8), im)


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_mutated_bags_end_in_messages_or_an_error(tmp_path, compression):
    """A bag is untrusted input: 300 random mutations of a valid file (bytes overwritten, runs zeroed or set to 0xff, truncations, a
    length field multiplied) are read to the end. Every one must give messages or a BagError — no crash, no hang, and no allocation
    beyond a few times the file (an address-space limit is in force while reading)."""
    import resource
    from cerberus_amd import rosbag as rb
    p = tmp_path / "ok.bag"
    rng = np.random.default_rng(20260926)
    with rb.BagWriter(p, compression=compression, chunk_threshold=2048) as w:
        for i in range(12):
            w.write(dict(kind=rb.KIND_IMU, topic="/imu", seq=i, secs=1, nsecs=i, linear_acceleration=rng.normal(size=3), angular_velocity=rng.normal(size=3)))
            w.write(dict(kind=rb.KIND_JOINT_STATE, topic="/leg", seq=i, secs=1, nsecs=i, position=rng.normal(size=16), velocity=rng.normal(size=16), effort=rng.normal(size=16)))
        w.write(dict(kind=rb.KIND_POINT_CLOUD, topic="/f", seq=0, secs=1, nsecs=50, points=rng.normal(size=(5, 3)).astype(np.float32),
                     channels=rng.normal(size=(6, 5)).astype(np.float32), channel_names=["id", "camera_id", "p_u", "p_v", "velocity_x", "velocity_y"]))
    raw = bytearray(p.read_bytes())
    n_ok = len(list(rb.BagReader(p)))
    assert n_ok == 25
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    outcomes = {"messages": 0, "error": 0}
    q = tmp_path / "mut.bag"
    try:
        for it in range(300):
            b = bytearray(raw)
            kind = it % 5
            at = int(rng.integers(0, len(b)))
            if kind == 0:
                for _ in range(int(rng.integers(1, 8))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:
                b[at:at + int(rng.integers(1, 64))] = b"\x00" * min(int(rng.integers(1, 64)), len(b) - at)
            elif kind == 2:
                n = min(int(rng.integers(1, 16)), len(b) - at)
                b[at:at + n] = b"\xff" * n
            elif kind == 3:
                b = b[:max(14, at)]
            else:
                at = min(at, len(b) - 4)
                v = struct.unpack_from("<I", b, at)[0]
                struct.pack_into("<I", b, at, (v * int(rng.integers(2, 1 << 16))) & 0xFFFFFFFF)
            q.write_bytes(bytes(b))
            # (the limit is set per read and lifted again: pytest itself may need more than the reader is allowed)
            used = 0
            try:
                with open("/proc/self/statm") as f:
                    used = int(f.read().split()[0]) * resource.getpagesize()
            except OSError:
                pass
            if used:
                resource.setrlimit(resource.RLIMIT_AS, (used + (512 << 20), hard))
            try:
                msgs = list(rb.BagReader(q))
                outcomes["messages"] += 1
                assert len(msgs) <= n_ok + 2
            except rb.BagError:
                outcomes["error"] += 1
            finally:
                resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
    finally:
        resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
    assert outcomes["messages"] + outcomes["error"] == 300 and outcomes["error"] > 30, outcomes
    if compression == "none":
        # text fields are bytes from the file too: a channel name that is not UTF-8 comes back with a replacement character
        b = bytearray(raw)
        at = bytes(b).rindex(b"velocity_x")   # (the cloud is the last message of the file)
        b[at + 3] = 0xa2
        q.write_bytes(bytes(b))
        names = [m["channel_names"] for m in rb.BagReader(q) if m["kind"] == rb.KIND_POINT_CLOUD]
        assert len(names) == 1 and names[0][4] == "vel\ufffdcity_x"


def test_connection_records_carry_the_dependencies_and_the_node_refuses_short_messages(tmp_path):
    """A connection's message_definition is what rosbag's Python API builds the classes from: the top-level text plus a "MSG:" section per
    dependency. And the node's logic names what it cannot use (a JointState without the four feet, a feature cloud without its six channels)
    instead of indexing past the end."""
    from cerberus_amd import rosbag as rb
    p = tmp_path / "defs.bag"
    with rb.BagWriter(p) as w:
        w.write(dict(kind=rb.KIND_IMU, topic="/imu", seq=0, secs=1, nsecs=0, linear_acceleration=np.zeros(3), angular_velocity=np.zeros(3)))
        w.write(dict(kind=rb.KIND_JOINT_STATE, topic="/leg", seq=0, secs=1, nsecs=0, position=np.zeros(16), velocity=np.zeros(16), effort=np.zeros(16)))
        w.write(dict(kind=rb.KIND_POINT_CLOUD, topic="/f", seq=0, secs=1, nsecs=0, points=np.zeros((1, 3), np.float32), channels=np.zeros((6, 1), np.float32),
                     channel_names=["id", "camera_id", "p_u", "p_v", "velocity_x", "velocity_y"]))
    raw = p.read_bytes()
    sep = b"=" * 80 + b"\n"
    for dep in (b"MSG: std_msgs/Header\nuint32 seq\ntime stamp\nstring frame_id\n", b"MSG: geometry_msgs/Quaternion\nfloat64 x\nfloat64 y\nfloat64 z\nfloat64 w\n",
                b"MSG: geometry_msgs/Vector3\nfloat64 x\nfloat64 y\nfloat64 z\n", b"MSG: geometry_msgs/Point32\nfloat32 x\nfloat32 y\nfloat32 z\n",
                b"MSG: sensor_msgs/ChannelFloat32\nstring name\nfloat32[] values\n"):
        assert sep + dep in raw, dep

    class Sink:
        def input_sample(self, t, s): pass
        def input_feature(self, t, ids, obs, stereo): return 0
        def process(self): return 0
    imu = dict(kind=rb.KIND_IMU, topic=rb.IMU_TOPIC, seq=0, secs=1, nsecs=0, linear_acceleration=np.zeros(3), angular_velocity=np.zeros(3))
    short = dict(kind=rb.KIND_JOINT_STATE, topic=rb.LEG_TOPIC, seq=0, secs=1, nsecs=0, position=np.zeros(12), velocity=np.zeros(12), effort=np.zeros(12))
    with pytest.raises(rb.BagError, match="12 joints"):
        rb.replay([imu, short], Sink())
    cloud = dict(kind=rb.KIND_POINT_CLOUD, topic=rb.FEATURE_TOPIC, seq=0, secs=1, nsecs=0, points=np.zeros((1, 3), np.float32), channels=np.zeros((2, 1), np.float32))
    with pytest.raises(rb.BagError, match="channels"):
        rb.replay([cloud], Sink())


def _records(raw, at, end):
    """top-level records of a bag file: (offset, header fields, data)"""
    out = []
    while at < end:
        hl = struct.unpack_from("<I", raw, at)[0]
        h, f, q = raw[at + 4:at + 4 + hl], {}, 0
        while q < len(h):
            n = struct.unpack_from("<I", h, q)[0]
            k, v = h[q + 4:q + 4 + n].split(b"=", 1)
            f[k.decode()] = v
            q += 4 + n
        dl = struct.unpack_from("<I", raw, at + 4 + hl)[0]
        out.append((at, f, raw[at + 8 + hl:at + 8 + hl + dl]))
        at += 8 + hl + dl
    return out


def _record(fields, data):
    h = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k.encode() + b"=" + v for k, v in fields.items())
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def test_compressed_chunks_against_independent_codecs(tmp_path):
    """bz2: the writer's chunks decompress with Python's bz2 to the chunks of the same bag written uncompressed, and a bag whose chunks were
    compressed by Python's bz2 (the file rebuilt record by record here) reads back message for message. lz4: the writer's chunks are LZ4
    frames as roslz4 writes them (magic 0x184D2204, version 01, independent blocks, content checksum) and decompress through liblz4's own
    frame API, called from here, to the uncompressed chunks."""
    import bz2
    import ctypes as C
    from cerberus_amd import rosbag as rb
    rng = np.random.default_rng(8)
    msgs = [dict(kind=rb.KIND_IMU, topic="/imu", seq=i, secs=5 + i // 50, nsecs=(i * 20000000) % 1000000000, frame_id="imu", linear_acceleration=rng.normal(size=3),
                 angular_velocity=rng.normal(size=3)) if i % 3 else
            dict(kind=rb.KIND_JOINT_STATE, topic="/leg", seq=i, secs=5 + i // 50, nsecs=(i * 20000000) % 1000000000, position=rng.normal(size=16), velocity=rng.normal(size=16),
                 effort=rng.normal(size=16)) for i in range(400)]
    raws = {}
    for comp in ("none", "bz2", "lz4"):
        with rb.BagWriter(tmp_path / (comp + ".bag"), chunk_threshold=8192, compression=comp) as w:
            for m in msgs:
                w.write(m)
        raws[comp] = (tmp_path / (comp + ".bag")).read_bytes()
    chunks = {c: [(f, d) for _, f, d in _records(raws[c], 13 + 4096, len(raws[c])) if f["op"] == b"\x05"] for c in raws}
    assert len(chunks["none"]) > 5 and len(chunks["bz2"]) == len(chunks["lz4"]) == len(chunks["none"])
    lz4 = C.CDLL("liblz4.so.1")
    lz4.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    lz4.LZ4F_decompress.restype = C.c_size_t
    lz4.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
    lz4.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
    for (f0, d0), (fb, db), (fl, dl) in zip(chunks["none"], chunks["bz2"], chunks["lz4"]):
        assert fb["compression"] == b"bz2" and fl["compression"] == b"lz4" and fb["size"] == fl["size"] == f0["size"] == struct.pack("<I", len(d0))
        assert bz2.decompress(db) == d0
        assert struct.unpack_from("<I", dl)[0] == 0x184D2204 and dl[4] >> 6 == 1 and dl[4] & 0x20 and dl[4] & 0x04 and not dl[4] & 0x08
        ctx = C.c_void_p()
        assert lz4.LZ4F_createDecompressionContext(C.byref(ctx), 100) == 0
        out, dn, sn = C.create_string_buffer(len(d0)), C.c_size_t(len(d0)), C.c_size_t(len(dl))
        assert lz4.LZ4F_decompress(ctx, out, C.byref(dn), dl, C.byref(sn), None) == 0 and dn.value == len(d0) and sn.value == len(dl)
        lz4.LZ4F_freeDecompressionContext(ctx)
        assert out.raw == d0
    # the reader on chunks it did not compress: the uncompressed file with every chunk recompressed by Python's bz2
    raw = raws["none"]
    body, index_pos = b"", None
    for at, f, d in _records(raw, 13 + 4096, len(raw)):
        if f["op"] == b"\x05":
            f = dict(f, compression=b"bz2")
            d = bz2.compress(d, 9)
        elif f["op"] == b"\x07" and index_pos is None:
            index_pos = 13 + 4096 + len(body)      # (the index section starts with the connection records)
        body += _record(f, d)
    (hat, hf, hd), = _records(raw, 13, 13 + 4096)
    hdr = _record(dict(hf, index_pos=struct.pack("<Q", index_pos)), hd)
    assert len(hdr) == 4096
    (tmp_path / "pybz2.bag").write_bytes(raw[:13] + hdr + body)
    back = list(rb.BagReader(tmp_path / "pybz2.bag"))
    assert len(back) == len(msgs)
    for a, b in zip(msgs, back):
        assert (a["kind"], a["seq"], a["secs"], a["nsecs"]) == (b["kind"], b["seq"], b["secs"], b["nsecs"])
        for key in ("linear_acceleration", "angular_velocity", "position", "velocity", "effort"):
            if key in a:
                np.testing.assert_array_equal(np.asarray(a[key]), b[key])


def test_stamps_are_ros_time():
    from cerberus_amd import rosbag as rb
    assert rb.to_stamp(12.0) == (12, 0) and rb.to_stamp(0.002) == (0, 2000000)
    assert rb.to_stamp(1.9999999996) == (2, 0)                     # the carry of fromSec
    for t in (0.0, 3.25, 1234.567891234, 0.998):
        s, ns = rb.to_stamp(t)
        assert abs(rb.to_sec(s, ns) - t) <= 0.5e-9 and 0 <= ns < 1000000000


def test_node_pairs_imu_with_leg_and_selects_the_contact_source(cfg):
    """sensor_callback (main.cpp:255-330): IMU and JointState messages of (nearly) the same stamp form one input; an unmatched head is dropped;
    CONTACT_SENSOR_TYPE 1 (and 0, in place of the absent Kalman filter) takes velocity[12 ..], type 2 effort[12 ..]."""
    from cerberus_amd import rosbag as rb, sequence
    got = []

    class FakeMp:
        def input_sample(self, t, s):
            got.append((t, s.copy()))

        def process(self):
            return 0

        def input_feature(self, *a):
            return 0

    def imu(t, v):
        s, ns = rb.to_stamp(t)
        return dict(kind=rb.KIND_IMU, topic=rb.IMU_TOPIC, seq=0, secs=s, nsecs=ns, linear_acceleration=np.full(3, v), angular_velocity=np.full(3, -v))

    def leg(t, v):
        s, ns = rb.to_stamp(t)
        return dict(kind=rb.KIND_JOINT_STATE, topic=rb.LEG_TOPIC, seq=0, secs=s, nsecs=ns, position=np.arange(16.0) + v, velocity=np.arange(16.0) * 2 + v,
                    effort=np.arange(16.0) * 3 + v)

    msgs = [imu(1.000, 1), leg(1.000, 1), imu(1.002, 2), imu(1.004, 3), leg(1.0040001, 3), leg(1.006, 4), imu(1.008, 5), leg(1.008, 5),
            dict(kind=rb.KIND_IMU, topic="/some/other/imu", seq=0, secs=1, nsecs=0, linear_acceleration=np.zeros(3), angular_velocity=np.zeros(3))]
    for ctype, src, scale in ((1, "velocity", 2), (0, "velocity", 2), (2, "effort", 3)):
        del got[:]
        cnt = rb.replay(msgs, FakeMp(), contact_sensor_type=ctype)
        assert cnt["pairs"] == 3 and cnt["dropped"] == 2 and cnt["other"] == 1
        assert [round(t, 6) for t, _ in got] == [1.0, 1.004, 1.008]
        for (t, s), v in zip(got, (1, 3, 5)):
            np.testing.assert_array_equal(s[1:4], np.full(3, v)); np.testing.assert_array_equal(s[4:7], np.full(3, -v))
            np.testing.assert_array_equal(s[7:19], np.arange(12.0) + v); np.testing.assert_array_equal(s[19:31], np.arange(12.0) * 2 + v)
            np.testing.assert_array_equal(s[31:35], np.arange(12.0, 16.0) * scale + v)


def test_stream_from_a_bag_gives_the_intervals_of_the_same_messages_from_memory(cfg, tmp_path):
    """The first images of a window need no device: a synthetic stream written as a bag (IMU, JointState, dummy image pairs, feature clouds)
    and read back drives vilo::MeasurementProcessor to exactly the intervals, dt and feature frames the in-memory messages give."""
    from cerberus_amd import rosbag as rb, sequence
    stream, frames = _frames(cfg, 8)
    t0 = frames[0]["header"] - len(frames[0]["samples"]) / 500.0
    msgs = rb.write_stream_bag(tmp_path / "s.bag", frames, t0, with_images=True, chunk_threshold=64 * 1024)
    tic, ric, _ = stream.extrinsics()
    runs = []
    for source in (msgs, rb.BagReader(tmp_path / "s.bag")):
        sw = sequence.SlidingWindow(None, cfg)
        sw.set_extrinsics(tic, ric, 0.0)
        mp = sequence.MeasurementProcessor(sw)
        log = []
        cnt = rb.replay(source, mp, on_image=lambda k, t: log.append((k, t, mp.last_interval(), sw.state()["feature_count"], sw.state()["Ps"].copy())))
        runs.append((cnt, log))
    (ca, la), (cb, lb) = runs
    assert ca == cb and ca["images"] == 16 and ca["clouds"] == 8 and ca["dropped"] == 0 and ca["processed"] >= 7
    assert len(la) == len(lb) >= 7
    for (ka, ta, ia, fa, pa), (kb, tb, ib, fb, pb) in zip(la, lb):
        assert (ka, ta, fa) == (kb, tb, fb)
        np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(pa, pb)
    # and the messages carry what the stream produced: the measurements exactly, dt to the nanosecond of the stamps
    want = frames[3]["samples"]
    got = la[3][2] if len(la[3][2]) == len(want) else la[3][2][-len(want):]
    np.testing.assert_array_equal(got[:, 1:], want[:, 1:])
    np.testing.assert_allclose(got[:, 0], want[:, 0], atol=2e-9)


@pytest.fixture(scope="module")
def ctx(cfg):
    from cerberus_amd import api
    c = api.Context(cfg, 0)
    yield c
    c.close()


@pytest.mark.gpu
def test_replay_from_a_bag_is_the_replay_from_memory_bit_for_bit(ctx, cfg, tmp_path):
    """25 images through solve + marginalisation on the device, once from the messages in memory and once from the bag they were written to:
    the same trajectory bit for bit. (The node starts like the reference's — at rest, attitude from the averaged accelerometer,
    estimator.cpp:524-544 — while the synthetic robot is already walking: the trajectory is compared with itself, not with the truth.)"""
    from cerberus_amd import rosbag as rb, sequence
    stream, frames = _frames(cfg, 25, seed=11)
    t0 = frames[0]["header"] - len(frames[0]["samples"]) / 500.0
    msgs = rb.write_stream_bag(tmp_path / "r.bag", frames, t0, with_images=False)
    tic, ric, td = stream.extrinsics()
    out = []
    for source in (msgs, rb.BagReader(tmp_path / "r.bag")):
        sw = sequence.SlidingWindow(ctx, cfg)
        sw.set_extrinsics(tic, ric, td)
        mp = sequence.MeasurementProcessor(sw)
        traj = []

        def on_image(k, t, sw=sw, traj=traj):
            st = sw.state()
            traj.append((t, st["frame_count"], st["n_optimizations"], st["Ps"].copy(), st["Vs"].copy(), st["Rho"].copy()))

        rb.replay(source, mp, on_image=on_image)
        out.append(traj)
    a, b = out
    assert len(a) == len(b) >= 24 and a[-1][2] >= 10
    for (ta, fa, na, pa, va, ra), (tb, fb, nb, pb, vb, rb_) in zip(a, b):
        assert (ta, fa, na) == (tb, fb, nb)
        np.testing.assert_array_equal(pa, pb); np.testing.assert_array_equal(va, vb); np.testing.assert_array_equal(ra, rb_)


@pytest.mark.gpu
@pytest.mark.parametrize("prior_form", ["factor", "eigen"])
def test_go1_parameter_replay_of_a_bag_matches_the_oracle_window_by_window(tmp_path, prior_form):
    """bench.py's replay configuration (BASELINE configs[4] stand-in: contact_sensor_type 2 foot forces, calf lengths estimated on line, the
    stream written as a bag and read back): every one of the first 30 windows the estimator solved on the GPU — dumped with the result — is
    solved again by the CPU oracle from the same dumped input (states 1e-8, same iteration count, cost 1e-7), and the prior the estimator
    carried into the next image is the oracle's marginalisation of the dumped result (normal equations 1e-5: cond(Amm))."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle_py as O
    from cerberus_amd import synth
    from test_sliding_window import _oracle_replay
    # (prior_form: what the marginalisation leaves as J0 — bench.py's replay lets it be the pivoted Cholesky factor, vilo_set_prior_form; the
    # oracle is handed the dumped prior either way and must reproduce the window's solve and the NEXT prior's normal equations)
    r = bench.replay_block(0, n_images=42, cpu_budget_s=0.0, keep_dir=str(tmp_path), prior_form=prior_form)
    assert "error" not in r, r
    dumps = os.path.join(str(tmp_path), "windows")
    n = len(os.listdir(dumps))
    assert n >= 30 and r["counters"]["dropped"] == 0 and r["counters"]["clouds"] == 42
    cfg5 = bench.go1_config(synth.default_config())
    assert cfg5.contact_sensor_type == 2
    _oracle_replay(O.config_from(cfg5), dumps, 30, tol_state=1e-8)
    assert r["rho_error_m"]["final"] < r["rho_error_m"]["at_start"]      # the calf lengths move towards the truth
