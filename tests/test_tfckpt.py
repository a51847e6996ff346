"""TensorFlow checkpoint files (tfckpt.py): the sorted-string-table index, the tensor bundle and the `checkpoint`
state file tf.train.Saver leaves behind (reference train.py:79-87,522-534,551-552).  No TensorFlow exists in this
image, so the format is pinned by its own invariants: RFC 3720's CRC-32C vectors, the table magic number, the byte
layout of hand-assembled blocks, and write -> read round trips."""
import os
import struct

import numpy as np
import pytest
import torch

from facet_graph_convolution_amd import tfckpt as T
from facet_graph_convolution_amd.net import FlatParams, param_spec


def test_crc32c_known_answers():
    # RFC 3720 B.4 test patterns + the customary check value
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert T.crc32c(b"") == 0
    # continuation across arbitrary (unaligned) split points equals one pass
    blob = np.random.RandomState(0).bytes(1000)
    for cut in (0, 1, 7, 8, 9, 511, 999, 1000):
        assert T.crc32c(blob[cut:], T.crc32c(blob[:cut])) == T.crc32c(blob)
    # masking as stored in the files is a bijection with the documented constant
    assert T.mask_crc(0) == 0xA282EAD8
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert T.unmask_crc(T.mask_crc(c)) == c


def test_table_layout_of_a_single_block(tmp_path):
    """Byte-for-byte: prefix compression, restart array, block trailer, footer."""
    p = str(tmp_path / "t.index")
    T.write_table(p, [(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")])
    raw = open(p, "rb").read()
    block = (bytes([0, 0, 1]) + b"H" +            # shared 0, unshared 0, value 1
             bytes([0, 2, 1]) + b"ab" + b"1" +     # first real key in full
             bytes([2, 1, 2]) + b"c" + b"22" +     # shares "ab" with its predecessor
             struct.pack("<II", 0, 1))             # one restart at offset 0
    assert raw[:len(block)] == block
    assert raw[len(block)] == 0                    # uncompressed
    assert struct.unpack_from("<I", raw, len(block) + 1)[0] == T.mask_crc(T.crc32c(block + b"\x00"))
    assert len(raw) >= T.FOOTER_BYTES and raw[-8:] == bytes.fromhex("57fb808b247547db")
    assert T.read_table(p) == [(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")]


def test_table_many_blocks_and_restarts(tmp_path):
    rs = np.random.RandomState(1)
    keys = sorted({("model/Level%d/Conv_%d/v%03d" % (rs.randint(3), rs.randint(4), i)).encode() for i in range(400)})
    items = [(b"", b"hdr")] + [(k, rs.bytes(rs.randint(0, 40))) for k in keys]
    for bs in (64, 1000, T.BLOCK_SIZE):        # many tiny blocks, a few, one
        p = str(tmp_path / ("t%d.index" % bs))
        T.write_table(p, items, block_size=bs)
        assert T.read_table(p) == items
    with pytest.raises(ValueError, match="strictly increasing"):
        T.write_table(str(tmp_path / "bad.index"), [(b"b", b""), (b"a", b"")])


def test_table_detects_corruption(tmp_path):
    p = str(tmp_path / "t.index")
    T.write_table(p, [(b"", b"H"), (b"k", b"value")])
    raw = bytearray(open(p, "rb").read())
    flipped = bytearray(raw)
    flipped[5] ^= 1
    open(p, "wb").write(flipped)
    with pytest.raises(ValueError, match="CRC-32C"):
        T.read_table(p)
    open(p, "wb").write(raw[:-1] + b"\x00")
    with pytest.raises(ValueError, match="magic"):
        T.read_table(p)
    open(p, "wb").write(raw[:20])
    with pytest.raises(ValueError, match="too short"):
        T.read_table(p)


def test_snappy_block_decoder():
    # literal "abcd", copy(offset 4, len 8) overlapping its own output, long literal via the 60-tag
    body = b"x" * 70
    stream = bytes([4 + 8 + 70]) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + \
        bytes([60 << 2, 69]) + body
    assert T._snappy_uncompress(stream) == b"abcd" + b"abcdabcd" + body
    with pytest.raises(ValueError):
        T._snappy_uncompress(bytes([5]) + bytes([3 << 2]) + b"abcd")       # length mismatch
    with pytest.raises(ValueError):
        T._snappy_uncompress(bytes([8]) + bytes([(4 << 2) | 2, 9, 0]))      # copy before the start


def test_bundle_entry_bytes():
    """BundleEntryProto wire bytes of a float [9,32,6] tensor at offset 300 (proto3: zero fields omitted)."""
    e = T._encode_entry(1, (9, 32, 6), 0, 300, 9 * 32 * 6 * 4, 0x01020304)
    d = T._decode_entry(e)
    assert d == {"dtype": 1, "shape": (9, 32, 6), "shard_id": 0, "offset": 300, "size": 6912, "crc32c": 0x01020304,
                 "slices": 0}
    assert e.startswith(bytes.fromhex("0801120c120208091202082012020806")) and e.endswith(b"\x35\x04\x03\x02\x01")
    scalar = T._encode_entry(3, (), 0, 0, 4, 7)
    assert scalar == bytes.fromhex("0803" "1200" "2804" "35" "07000000")
    assert T._decode_entry(scalar)["shape"] == ()
    assert T._decode_header(T._encode_header(1)) == {"num_shards": 1, "endianness": 0}


def test_bundle_round_trip_and_checks(tmp_path):
    rs = np.random.RandomState(2)
    tensors = {"model/Level0/Conv/weight": rs.normal(size=(9, 32, 6)).astype(np.float32),
               "model/Level0/Conv/bias": rs.normal(size=(32,)).astype(np.float32),
               "beta1_power": np.float32(0.9), "Variable": np.int32(1234),
               "d": rs.normal(size=(3, 0)).astype(np.float64), "i64": np.arange(5, dtype=np.int64)}
    prefix = str(tmp_path / "sub" / "net-1234")
    T.write_bundle(prefix, tensors)
    assert sorted(os.listdir(str(tmp_path / "sub"))) == ["net-1234.data-00000-of-00001", "net-1234.index"]
    got = T.read_bundle(prefix)
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.shape(v) and np.array_equal(got[k], v)
    # data file = tensors back to back in key order
    ents, hdr = T.bundle_entries(prefix)
    assert hdr["num_shards"] == 1
    off = 0
    for k in sorted(tensors, key=lambda s: s.encode()):
        assert ents[k]["offset"] == off and ents[k]["size"] == np.asarray(tensors[k]).nbytes
        off += ents[k]["size"]
    assert os.path.getsize(prefix + ".data-00000-of-00001") == off
    assert list(T.read_bundle(prefix, names={"Variable"})) == ["Variable"]
    with pytest.raises(KeyError, match="holds no variable"):
        T.read_bundle(prefix, names={"nope"})
    # a flipped data byte is caught by the per-tensor checksum
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[10] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(raw)
    with pytest.raises(ValueError, match="fails its CRC-32C"):
        T.read_bundle(prefix)
    assert T.read_bundle(prefix, verify=False)
    open(prefix + ".data-00000-of-00001", "wb").write(raw[:100])
    with pytest.raises(ValueError, match="ends inside"):
        T.read_bundle(prefix, verify=False)


def test_checkpoint_state_file(tmp_path):
    d = str(tmp_path)
    assert T.get_checkpoint_state(d) is None
    for step in range(0, 700, 100):
        T.write_bundle(os.path.join(d, "net-%d" % step), {"v": np.float32(step)})
        T.update_checkpoint_state(os.path.join(d, "net-%d" % step))
    st = T.get_checkpoint_state(d)
    assert st.model_checkpoint_path == os.path.join(d, "net-600")
    assert [os.path.basename(p) for p in st.all_model_checkpoint_paths] == ["net-%d" % s for s in range(200, 700, 100)]
    assert not os.path.exists(os.path.join(d, "net-0.index")) and os.path.exists(os.path.join(d, "net-200.index"))
    assert open(os.path.join(d, "checkpoint")).readline() == 'model_checkpoint_path: "net-600"\n'
    # every spelling of "that checkpoint" resolves to the prefix
    want = os.path.join(d, "net-600")
    for spelling in (d, want, want + ".index", want + ".data-00000-of-00001"):
        assert T.resolve_prefix(spelling) == want and T.is_tf_checkpoint(spelling)
    assert not T.is_tf_checkpoint(os.path.join(d, "other"))
    # an absolute path written by TensorFlow is kept as is
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "/abs/net-7"\n')
    assert T.get_checkpoint_state(d).model_checkpoint_path == "/abs/net-7"


class _HostNet:
    def __init__(self, multi_scale, seed):
        self.multi_scale = multi_scale
        self.params = FlatParams(param_spec(multi_scale), "cpu")
        self.params.init_random(seed)


def test_variable_name_table():
    names = T.variable_names(False)
    assert len(names) == 44 == len(param_spec(False)) and len(set(names)) == 44
    assert names[:5] == ["model/Level0/Conv/" + v for v in ("weight", "bias", "assignment", "assignment_1", "assignment_2")]
    assert names[15] == "model/Level2/Conv_1/weight" and names[20] == "model/Level1_1/Conv/weight"
    assert names[-4:] == ["model/Level0_1/MLP/weight", "model/Level0_1/MLP/bias", "model/Level0_1/MLP_1/weight",
                          "model/Level0_1/MLP_1/bias"]
    ms = T.variable_names(True)
    assert len(ms) == 52 == len(param_spec(True)) and ms[20:24] == [
        "model/Level2/MLP/weight", "model/Level2/MLP/bias", "model/Level2/MLP_1/weight", "model/Level2/MLP_1/bias"]
    # kinds line up with the creation-order spec
    for n, (kind, _) in zip(ms, param_spec(True)):
        assert n.rsplit("/", 1)[1].split("_")[0] == kind


@pytest.mark.parametrize("multi_scale", [False, True])
def test_network_save_restore(tmp_path, multi_scale):
    net = _HostNet(multi_scale, 3)
    P = net.params
    rs = np.random.RandomState(4)
    P.m.copy_(torch.from_numpy(rs.normal(size=P.total).astype(np.float32)))
    P.v.copy_(torch.from_numpy(rs.uniform(size=P.total).astype(np.float32)))
    P.step = 4321
    prefix = T.save_network(str(tmp_path / "net"), net, global_step=4321)
    assert prefix.endswith("net-4321")
    ents, _ = T.bundle_entries(prefix)
    nvar = len(P.spec)
    assert len(ents) == 3 * nvar + 3 and ents["model/Level0/Conv/weight"]["shape"] == (9, 32, 6)
    assert ents["model/Level0_1/MLP/weight/Adam_1"]["shape"] == (32, 1024) and ents["Variable"]["dtype"] == 3
    other = _HostNet(multi_scale, 9)
    assert T.load_network(str(tmp_path), other) == 4321          # through the `checkpoint` state file
    for a, b in zip(P.values, other.params.values):
        assert torch.equal(a, b)
    for k in ("m", "v"):
        for o, (_, s) in zip(P.offsets, P.spec):
            n = int(np.prod(s))
            assert torch.equal(getattr(P, k)[o:o + n], getattr(other.params, k)[o:o + n])
    assert other.params.step == 4321
    # the other architecture is refused: the single-scale file lacks the heads, the multi-scale one has the same 44
    # names and more, so a single-scale network can read it
    if not multi_scale:
        with pytest.raises(KeyError, match="lacks 8 of the network's 52"):
            T.load_network(prefix, _HostNet(True, 0))
    else:
        single = _HostNet(False, 0)
        T.load_network(prefix, single)
        assert torch.equal(single.params.values[0], P.values[0]) and torch.equal(single.params.values[-1], P.values[-1])


def test_weights_only_checkpoint_and_name_map(tmp_path):
    """What the reference's inference graph would save (no optimizer), under differently spelt scopes."""
    net = _HostNet(False, 5)
    names = T.variable_names(False)
    tensors = {n.replace("model/", "net/"): v.numpy() for n, v in zip(names, net.params.values)}
    prefix = str(tmp_path / "w-77")
    T.write_bundle(prefix, tensors)
    other = _HostNet(False, 6)
    with pytest.raises(KeyError, match="lacks 44 of the network's 44"):
        T.load_network(prefix, other)
    assert T.load_network(prefix, other, name_map={n: n.replace("model/", "net/") for n in names}) == 77
    assert torch.equal(other.params.theta, net.params.theta) and other.params.step == 0
    assert float(other.params.m.abs().max()) == 0.0
    with pytest.raises(KeyError, match="Adam"):
        T.load_network(prefix, other, name_map={n: n.replace("model/", "net/") for n in names}, strict_optimizer=True)
    bad = dict(tensors)
    bad["net/Level0/Conv/weight"] = np.zeros((9, 32, 3), np.float32)
    T.write_bundle(prefix, bad)
    with pytest.raises(ValueError, match="has shape"):
        T.load_network(prefix, other, name_map={n: n.replace("model/", "net/") for n in names})


def test_step_recovery_without_a_counter(tmp_path):
    net = _HostNet(False, 1)
    net.params.step = 57
    prefix = T.save_network(str(tmp_path / "n"), net, global_step=900)
    tensors = T.read_bundle(prefix)
    del tensors["Variable"]
    T.write_bundle(prefix, tensors)
    other = _HostNet(False, 2)
    assert T.load_network(prefix, other) == 900 and other.params.step == 57      # from beta1_power = 0.9^58
    tensors["beta1_power"] = np.float32(0.0)                                      # underflowed after a long run
    T.write_bundle(prefix, tensors)
    assert T.load_network(prefix, other) == 900 and other.params.step == 900      # from the file name
