"""TensorFlow checkpoint files, read and written without TensorFlow.

The reference keeps its networks with ``tf.train.Saver`` (train.py:79-87 restore for inference, :522-534 restore
for training, :551-552,626 save every SAVEITER iterations): ``<NETWORK_PATH><NET_NAME>-<step>.index`` +
``.data-00000-of-00001`` (a "tensor bundle") and a ``checkpoint`` state file next to them.  This module restates
those published file formats so a network trained by the reference can drive the HIP kernels and the other way
round:

* ``checkpoint``: text ``CheckpointState`` - ``model_checkpoint_path: "..."`` / ``all_model_checkpoint_paths``;
* ``.index``: a LevelDB-style sorted string table (prefix-compressed blocks with restart arrays, a 5-byte trailer
  = compression byte + masked CRC-32C per block, index block, 48-byte footer ending in 0xdb4775248b80fb57).  Key ""
  holds a ``BundleHeaderProto``; every other key is a variable name and holds a ``BundleEntryProto``
  (dtype, shape, shard_id, offset, size, masked crc32c);
* ``.data-SSSSS-of-NNNNN``: the raw little-endian tensor bytes at those offsets.

Variable names follow TF1 scoping of the reference's graph (train.py:71 ``variable_scope("model")``; model.py:853-
941 ``Level0/1/2`` entered twice, the second time as name scope ``LevelN_1``; model.py:428,764 ``Conv`` / ``MLP``,
uniquified per level; model.py:31-44 ``weight`` / ``bias`` / ``assignment`` [u], ``assignment_1`` [c],
``assignment_2`` [v]).  No TensorFlow and no reference checkpoint exist in this image, so that table and the
format are pinned only by the format's own invariants (magic number, RFC 3720 CRC vectors, round trips):
parity unpinned for this module - ``read_bundle`` lists what a file really holds and ``name_map=`` overrides the
table if a real checkpoint disagrees.
"""
import os
import re
import struct

import numpy as np

from . import _lib

TABLE_MAGIC = 0xDB4775248B80FB57
BLOCK_TRAILER = 5
FOOTER_BYTES = 48
RESTART_INTERVAL = 16
BLOCK_SIZE = 262144
MAX_TO_KEEP = 5  # tf.train.Saver default

# tensorflow DataType enum <-> numpy
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def crc32c(data, crc=0):
    data = bytes(data)
    return int(_lib.lib().fgc_crc32c(crc, data, len(data)))


def _crc32c_array(a):
    a = np.ascontiguousarray(a)
    return int(_lib.lib().fgc_crc32c(0, _lib.C.c_void_p(a.ctypes.data), a.nbytes))


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------
# protobuf wire format (the three messages the bundle uses)
# ---------------------------------------------------------------------------------------------------
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint longer than 64 bits")


def _fields(buf):
    """(field number, wire type, value) triples of one message; value is int or bytes."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            val, pos = bytes(buf[pos:pos + n]), pos + n
            if len(val) != n:
                raise ValueError("truncated length-delimited field")
        elif wt == 5:
            val, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _encode_header(num_shards=1):
    out = bytearray()
    out += b"\x08"
    _put_varint(out, num_shards)          # num_shards = 1; endianness LITTLE = 0 is the default and is omitted
    out += b"\x1a\x02\x08\x01"            # version { producer: 1 }
    return bytes(out)


def _decode_header(buf):
    h = {"num_shards": 0, "endianness": 0}
    for num, _, val in _fields(buf):
        if num == 1:
            h["num_shards"] = val
        elif num == 2:
            h["endianness"] = val
    return h


def _encode_entry(dtype_id, shape, shard_id, offset, size, crc_masked):
    out = bytearray()
    out += b"\x08"
    _put_varint(out, dtype_id)
    sh = bytearray()
    for d in shape:
        dim = bytearray(b"\x08")
        _put_varint(dim, int(d))
        sh += b"\x12"
        _put_varint(sh, len(dim))
        sh += dim
    out += b"\x12"
    _put_varint(out, len(sh))
    out += sh
    if shard_id:
        out += b"\x18"
        _put_varint(out, shard_id)
    if offset:
        out += b"\x20"
        _put_varint(out, offset)
    if size:
        out += b"\x28"
        _put_varint(out, size)
    out += b"\x35" + struct.pack("<I", crc_masked)
    return bytes(out)


def _decode_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for num, _, val in _fields(buf):
        if num == 1:
            e["dtype"] = val
        elif num == 2:
            dims = []
            for n2, _, v2 in _fields(val):
                if n2 == 2:
                    size = 0
                    for n3, _, v3 in _fields(v2):
                        if n3 == 1:
                            size = _signed(v3)
                    dims.append(size)
                elif n2 == 3 and v2:
                    raise ValueError("tensor of unknown rank in checkpoint")
            e["shape"] = tuple(dims)
        elif num == 3:
            e["shard_id"] = val
        elif num == 4:
            e["offset"] = val
        elif num == 5:
            e["size"] = val
        elif num == 6:
            e["crc32c"] = val
        elif num == 7:
            e["slices"] += 1
    return e


# ---------------------------------------------------------------------------------------------------
# sorted string table
# ---------------------------------------------------------------------------------------------------
def _snappy_uncompress(buf):
    """Raw snappy block (the table's compression type 1).  TensorFlow writes bundle indexes uncompressed; this is
    only here so a recompressed file is still readable."""
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):        # byte-wise: the copy may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy block decodes to %d bytes, header says %d" % (len(out), n))
    return bytes(out)


def _read_block(data, offset, size, what):
    end = offset + size + BLOCK_TRAILER
    if end > len(data):
        raise ValueError("%s block [%d,+%d) runs past the end of the file" % (what, offset, size))
    body, ctype = data[offset:offset + size], data[offset + size]
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
        raise ValueError("%s block at offset %d fails its CRC-32C" % (what, offset))
    if ctype == 1:
        body = _snappy_uncompress(body)
    elif ctype != 0:
        raise ValueError("%s block has unknown compression type %d" % (what, ctype))
    return body


def _block_entries(block):
    if len(block) < 4:
        raise ValueError("table block shorter than its restart count")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestarts
    if limit < 0:
        raise ValueError("table block restart array larger than the block")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        nonshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + nonshared + vlen > limit:
            raise ValueError("corrupt table block entry")
        key = key[:shared] + block[pos:pos + nonshared]
        pos += nonshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _decode_handle(buf, pos=0):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def read_table(path):
    """All (key, value) pairs of a table file in key order."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < FOOTER_BYTES:
        raise ValueError("%s: too short to be a table file" % path)
    footer = data[-FOOTER_BYTES:]
    magic = struct.unpack_from("<Q", footer, FOOTER_BYTES - 8)[0]
    if magic != TABLE_MAGIC:
        raise ValueError("%s: bad table magic %#x (not a TensorFlow V2 checkpoint index)" % (path, magic))
    _, _, pos = _decode_handle(footer)               # metaindex handle (unused by the bundle)
    ioff, isize, _ = _decode_handle(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, "index")):
        boff, bsize, _ = _decode_handle(handle)
        out.extend(_block_entries(_read_block(data, boff, bsize, "data")))
    for (k0, _), (k1, _) in zip(out, out[1:]):
        if not k0 < k1:
            raise ValueError("%s: keys out of order (%r then %r)" % (path, k0, k1))
    return out


class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + \
            struct.pack("<I", len(self.restarts))


def _encode_handle(off, size):
    out = bytearray()
    _put_varint(out, off)
    _put_varint(out, size)
    return bytes(out)


def write_table(path, items, block_size=BLOCK_SIZE):
    """items: (key bytes, value bytes) in strictly increasing key order; uncompressed blocks as BundleWriter does."""
    out = bytearray()
    index = _BlockBuilder()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    cur, last = _BlockBuilder(), None
    for key, value in items:
        if last is not None and not last < key:
            raise ValueError("table keys must be strictly increasing (%r then %r)" % (last, key))
        cur.add(key, value)
        last = key
        if cur.size() >= block_size:
            index.add(last, _encode_handle(*emit(cur.finish())))
            cur = _BlockBuilder()
    if cur.count:
        index.add(last, _encode_handle(*emit(cur.finish())))
    meta = _encode_handle(*emit(_BlockBuilder().finish()))
    idx = _encode_handle(*emit(index.finish()))
    footer = (meta + idx).ljust(FOOTER_BYTES - 8, b"\x00") + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(out)
    os.replace(tmp, path)


# ---------------------------------------------------------------------------------------------------
# tensor bundle
# ---------------------------------------------------------------------------------------------------
def _shard_name(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def bundle_entries(prefix):
    """{name: entry dict} of `<prefix>.index` plus the header dict."""
    items = read_table(prefix + ".index")
    if not items or items[0][0] != b"":
        raise ValueError("%s.index: no bundle header entry" % prefix)
    header = _decode_header(items[0][1])
    if header["endianness"] != 0:
        raise ValueError("%s: big-endian bundle" % prefix)
    return {k.decode("utf-8"): _decode_entry(v) for k, v in items[1:]}, header


def read_bundle(prefix, names=None, verify=True):
    """{variable name: numpy array} of the checkpoint `<prefix>.index` + `<prefix>.data-*`."""
    entries, header = bundle_entries(prefix)
    files, out = {}, {}
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e["slices"]:
                raise ValueError("%s: variable %s is saved in slices (partitioned variable)" % (prefix, name))
            if e["dtype"] not in _DTYPES:
                raise ValueError("%s: variable %s has unsupported dtype enum %d" % (prefix, name, e["dtype"]))
            dt = np.dtype(_DTYPES[e["dtype"]])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if count * dt.itemsize != e["size"]:
                raise ValueError("%s: variable %s: shape %s x %s needs %d bytes, entry says %d" %
                                 (prefix, name, e["shape"], dt, count * dt.itemsize, e["size"]))
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open(_shard_name(prefix, sid, max(header["num_shards"], 1)), "rb")
            fh = files[sid]
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError("%s: data file ends inside variable %s" % (prefix, name))
            arr = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
            if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != _crc32c_array(arr):
                raise ValueError("%s: variable %s fails its CRC-32C" % (prefix, name))
            out[name] = arr
    finally:
        for fh in files.values():
            fh.close()
    if names is not None:
        missing = [n for n in names if n not in out]
        if missing:
            raise KeyError("%s holds no variable(s) %s; it holds %s" % (prefix, missing, sorted(entries)))
    return out


def write_bundle(prefix, tensors, block_size=BLOCK_SIZE):
    """Writes {name: array} as `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, names sorted)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    items, offset = [(b"", _encode_header(1))], 0
    data_path = _shard_name(prefix, 0, 1)
    tmp = data_path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as fh:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            if not name:
                raise ValueError("empty variable name")
            arr = np.asarray(tensors[name])
            if not arr.flags.c_contiguous:
                arr = np.ascontiguousarray(arr)
            if arr.dtype.byteorder == ">":
                arr = arr.astype(arr.dtype.newbyteorder("<"))
            if arr.dtype not in _DTYPE_IDS:
                raise ValueError("variable %s: dtype %s has no TensorFlow checkpoint type here" % (name, arr.dtype))
            fh.write(arr.tobytes())
            items.append((name.encode("utf-8"), _encode_entry(_DTYPE_IDS[arr.dtype], arr.shape, 0, offset, arr.nbytes,
                                                              mask_crc(_crc32c_array(arr)))))
            offset += arr.nbytes
    os.replace(tmp, data_path)
    write_table(prefix + ".index", items, block_size=block_size)


# ---------------------------------------------------------------------------------------------------
# `checkpoint` state file
# ---------------------------------------------------------------------------------------------------
_STATE_LINE = re.compile(r'^\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"((?:[^"\\]|\\.)*)"\s*$')


class CheckpointState:
    def __init__(self, model_checkpoint_path=None, all_model_checkpoint_paths=()):
        self.model_checkpoint_path = model_checkpoint_path
        self.all_model_checkpoint_paths = list(all_model_checkpoint_paths)


def get_checkpoint_state(checkpoint_dir):
    """tf.train.get_checkpoint_state: None without a `checkpoint` file; relative paths resolved against the dir."""
    path = os.path.join(checkpoint_dir or ".", "checkpoint")
    if not os.path.exists(path):
        return None
    st = CheckpointState()

    def resolve(p):
        p = p.encode("utf-8").decode("unicode_escape")
        return p if os.path.isabs(p) else os.path.join(checkpoint_dir or ".", p)

    with open(path) as fh:
        for line in fh:
            m = _STATE_LINE.match(line)
            if not m:
                continue
            if m.group(1) == "model_checkpoint_path":
                st.model_checkpoint_path = resolve(m.group(2))
            else:
                st.all_model_checkpoint_paths.append(resolve(m.group(2)))
    return st


def update_checkpoint_state(prefix, max_to_keep=MAX_TO_KEEP):
    """Records `prefix` as the latest checkpoint of its directory and deletes the ones beyond max_to_keep, as
    tf.train.Saver.save does."""
    d = os.path.dirname(os.path.abspath(prefix))
    base = os.path.basename(prefix)
    st = get_checkpoint_state(d)
    kept = [os.path.basename(p) for p in (st.all_model_checkpoint_paths if st else [])
            if os.path.dirname(os.path.abspath(p)) == d]
    kept = [p for p in kept if p != base] + [base]
    while max_to_keep and len(kept) > max_to_keep:
        old = os.path.join(d, kept.pop(0))
        for f in (old + ".index", _shard_name(old, 0, 1)):
            if os.path.exists(f):
                os.remove(f)
    tmp = os.path.join(d, "checkpoint.tmp%d" % os.getpid())
    with open(tmp, "w") as fh:
        fh.write('model_checkpoint_path: "%s"\n' % base)
        for p in kept:
            fh.write('all_model_checkpoint_paths: "%s"\n' % p)
    os.replace(tmp, os.path.join(d, "checkpoint"))


# ---------------------------------------------------------------------------------------------------
# the reference network's variables
# ---------------------------------------------------------------------------------------------------
GLOBAL_STEP_NAME = "Variable"            # train.py:429  batch = tf.Variable(0, trainable=False)
ADAM_BETA1, ADAM_BETA2 = 0.9, 0.999      # tf.train.AdamOptimizer() defaults (train.py:520)


def variable_names(multi_scale=False, scope="model"):
    """TF names of the network's variables in creation order = net.param_spec order (see module docstring)."""
    conv = ["weight", "bias", "assignment", "assignment_1", "assignment_2"]
    lin = ["weight", "bias"]
    layers = [("Level0", "Conv", conv), ("Level1", "Conv", conv), ("Level2", "Conv", conv), ("Level2", "Conv_1", conv)]
    if multi_scale:
        layers += [("Level2", "MLP", lin), ("Level2", "MLP_1", lin)]
    layers += [("Level1_1", "Conv", conv), ("Level1_1", "Conv_1", conv)]
    if multi_scale:
        layers += [("Level1_1", "MLP", lin), ("Level1_1", "MLP_1", lin)]
    layers += [("Level0_1", "Conv", conv), ("Level0_1", "Conv_1", conv), ("Level0_1", "MLP", lin),
               ("Level0_1", "MLP_1", lin)]
    pre = scope + "/" if scope else ""
    return ["%s%s/%s/%s" % (pre, lvl, op, v) for lvl, op, vs in layers for v in vs]


def resolve_prefix(path):
    """`path` may be a checkpoint prefix, one of its files, or a directory holding a `checkpoint` state file."""
    if os.path.isdir(path):
        st = get_checkpoint_state(path)
        if st is None or not st.model_checkpoint_path:
            raise FileNotFoundError("%s: no `checkpoint` state file" % path)
        return st.model_checkpoint_path
    for suffix in (".index",):
        if path.endswith(suffix):
            return path[:-len(suffix)]
    m = re.match(r"^(.*)\.data-\d{5}-of-\d{5}$", path)
    return m.group(1) if m else path


def is_tf_checkpoint(path):
    if os.path.isdir(path):
        return os.path.exists(os.path.join(path, "checkpoint"))
    return os.path.exists(resolve_prefix(path) + ".index")


def save_network(prefix, net, global_step=None, state_file=True):
    """What saver.save(sess, prefix, global_step=...) leaves behind for the reference's training graph: weights,
    Adam slots `<var>/Adam` (m) and `<var>/Adam_1` (v), beta1_power / beta2_power and the step counter."""
    P = net.params
    if global_step is not None:
        prefix = "%s-%d" % (prefix, global_step)
    names = variable_names(net.multi_scale)
    if len(names) != len(P.spec):
        raise RuntimeError("network has %d variables, the name table %d" % (len(P.spec), len(names)))
    host = {k: getattr(P, k).detach().cpu().numpy() for k in ("theta", "m", "v")}
    tensors = {}
    for name, off, (_, shape) in zip(names, P.offsets, P.spec):
        n = int(np.prod(shape))
        tensors[name] = host["theta"][off:off + n].reshape(shape)
        tensors[name + "/Adam"] = host["m"][off:off + n].reshape(shape)
        tensors[name + "/Adam_1"] = host["v"][off:off + n].reshape(shape)
    # AdamOptimizer keeps beta^(t+1) after t updates (initialised to beta, multiplied once per step)
    tensors["beta1_power"] = np.float32(np.float32(ADAM_BETA1) ** (P.step + 1))
    tensors["beta2_power"] = np.float32(np.float32(ADAM_BETA2) ** (P.step + 1))
    tensors[GLOBAL_STEP_NAME] = np.int32(P.step)
    write_bundle(prefix, tensors)
    if state_file:
        update_checkpoint_state(prefix)
    return prefix


def load_network(path, net, name_map=None, strict_optimizer=False):
    """saver.restore for a FacetDenoiser.  Weights are required (all of them, shapes checked); Adam slots and the
    step counter are taken when present (a checkpoint written by the reference's inference graph has none).
    name_map: {expected TF name: name in the file} overrides.  Returns the step parsed from `<name>-<step>`."""
    prefix = resolve_prefix(path)
    entries, _ = bundle_entries(prefix)
    P = net.params
    names = variable_names(net.multi_scale)
    if name_map:
        names = [name_map.get(n, n) for n in names]
    missing = [n for n in names if n not in entries]
    if missing:
        raise KeyError("%s lacks %d of the network's %d variables (first: %s); it holds: %s" %
                       (prefix, len(missing), len(names), missing[0], ", ".join(sorted(entries))))
    for n, (_, shape) in zip(names, P.spec):
        if tuple(entries[n]["shape"]) != tuple(shape):
            raise ValueError("%s: variable %s has shape %s, the network needs %s" %
                             (prefix, n, tuple(entries[n]["shape"]), tuple(shape)))
    slots = [n + s for n in names for s in ("/Adam", "/Adam_1")]
    have_slots = all(s in entries for s in slots)
    if strict_optimizer and not have_slots:
        raise KeyError("%s holds no Adam slots" % prefix)
    extra = [n for n in ("beta1_power", GLOBAL_STEP_NAME) if n in entries]
    got = read_bundle(prefix, names=set(names) | (set(slots) if have_slots else set()) | set(extra))
    host = {k: np.zeros(P.total, dtype=np.float32) for k in ("theta", "m", "v")}
    for n, off, (_, shape) in zip(names, P.offsets, P.spec):
        cnt = int(np.prod(shape))
        host["theta"][off:off + cnt] = got[n].astype(np.float32).reshape(-1)
        if have_slots:
            host["m"][off:off + cnt] = got[n + "/Adam"].astype(np.float32).reshape(-1)
            host["v"][off:off + cnt] = got[n + "/Adam_1"].astype(np.float32).reshape(-1)
    import torch
    for k in ("theta", "m", "v"):
        getattr(P, k).copy_(torch.from_numpy(host[k]))
    m = re.match(r"^.*-(\d+)$", os.path.basename(prefix))
    file_step = int(m.group(1)) if m else 0
    if have_slots:
        # the step counter drives Adam's bias correction: the saved counter if there is one, else beta1_power
        # (= 0.9^(t+1), which underflows fp32 after a few thousand steps), else the number in the file name
        step = file_step
        b1p = float(got["beta1_power"]) if "beta1_power" in got else 0.0
        if 1e-30 < b1p < 1.0:
            step = max(int(round(np.log(b1p) / np.log(ADAM_BETA1))) - 1, 0)
        gs = got.get(GLOBAL_STEP_NAME)
        if gs is not None and np.ndim(gs) == 0 and np.issubdtype(np.asarray(gs).dtype, np.integer):
            step = int(gs)
        P.step = step
    else:
        P.step = 0
    return file_step if m else P.step
