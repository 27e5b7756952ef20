"""TensorFlow checkpoint (TensorBundle V2) reader / writer without TensorFlow.

Replaces what the reference gets from `tf.train.NewCheckpointReader` / `tf.train.Saver.restore`
(Data_utils/weights_utils.py:29-37, Stereo_Online_Adaptation.py:150-153): the released MADNet / DispNet weights
are TF checkpoints `<prefix>.index` + `<prefix>.data-00000-of-0000N` (README.MD:47).

Format (tensorflow/core/util/tensor_bundle + lib/io/table = the LevelDB table format):
  <prefix>.index   SSTable: [data blocks][metaindex block][index block][48-byte footer]
                   block  = entries (shared|non_shared|value_len varint32, key delta, value) + uint32 restarts[] +
                            uint32 num_restarts, followed by a 5-byte trailer (compression type, masked crc32c)
                   footer = BlockHandle(metaindex) BlockHandle(index) (varint64 offset,size), zero padded to 40 bytes,
                            magic 0xdb4775248b80fb57 (little endian)
                   key "" -> BundleHeaderProto{num_shards=1, endianness=2, version=3}
                   key <variable name> -> BundleEntryProto{dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6(fixed32)}
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at (offset, size)
No real TF checkpoint exists in this environment (no network): the reader is validated by round trips through the
writer below and by hand-checked byte layouts in tests/test_tf_checkpoint.py -- treat it as unverified against
TensorFlow itself until a released checkpoint has been read with it.
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT_NUMPY = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
            17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
NUMPY_DT = {np.dtype(v): k for k, v in DT_NUMPY.items()}

# ---- crc32c (Castagnoli), masked the LevelDB way ------------------------------------------------------------
_T = None


def _table():
    global _T
    if _T is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            t.append(c)
        _T = t
    return _T


def crc32c(data, crc=0):
    """Table driven; large buffers go through libmadnet_hip's mh_crc32c when the library is loadable."""
    data = bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data
    if len(data) >= 1 << 16:
        try:
            from madnet_hip import _ffi
            import ctypes as C
            dll = C.CDLL(_ffi.LIB_PATH)
            dll.mh_crc32c.restype = C.c_uint32
            dll.mh_crc32c.argtypes = [C.c_char_p, C.c_int64, C.c_uint32]
            return int(dll.mh_crc32c(bytes(data), len(data), crc))
        except Exception:
            pass
    t = _table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / minimal protobuf ------------------------------------------------------------------------------
def _get_varint(buf, pos):
    r, s = 0, 0
    while True:
        b = buf[pos]; pos += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, pos
        s += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field, wire_type, value) ; value = int (varint / fixed) or bytes (length delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        f, w = tag >> 3, tag & 7
        if w == 0:
            v, pos = _get_varint(buf, pos)
        elif w == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif w == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif w == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % w)
        out.append((f, w, v))
    return out


def _field(tag, wire, payload):
    return _put_varint((tag << 3) | wire) + payload


def _parse_entry(val):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, w, v in _parse_proto(val):
        if f == 1: e["dtype"] = v
        elif f == 2:
            for f2, w2, v2 in _parse_proto(v):
                if f2 == 2:                       # TensorShapeProto.dim
                    size = 0
                    for f3, w3, v3 in _parse_proto(v2):
                        if f3 == 1: size = v3
                    e["shape"].append(size)
                elif f2 == 3 and v2:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = v
        elif f == 7: e["slices"] += 1
    return e


# ---- SSTable ---------------------------------------------------------------------------------------------------
def _read_block(buf, off, size, verify=True):
    data = buf[off:off + size]
    ctype = buf[off + size]
    if verify:
        stored = struct.unpack_from("<I", buf, off + size + 1)[0]
        if mask_crc(crc32c(bytes(data) + bytes([ctype]))) != stored:
            raise ValueError("checkpoint index: block checksum mismatch at offset %d" % off)
    if ctype != 0:
        raise NotImplementedError("compressed (type %d) index blocks are not supported (TensorBundle writes them uncompressed)" % ctype)
    nrestart = struct.unpack_from("<I", data, len(data) - 4)[0]
    end = len(data) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(data, pos)
        non_shared, pos = _get_varint(data, pos)
        vlen, pos = _get_varint(data, pos)
        key = key[:shared] + bytes(data[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(data[pos:pos + vlen]))); pos += vlen
    return out


def _read_table(path):
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("%s is not a TensorBundle index (bad SSTable magic)" % path)
    foot = buf[len(buf) - 48:]
    pos = 0
    _, pos = _get_varint(foot, pos); _, pos = _get_varint(foot, pos)          # metaindex handle
    ioff, pos = _get_varint(foot, pos); isz, pos = _get_varint(foot, pos)
    entries = []
    for _, handle in _read_block(buf, ioff, isz):
        boff, p = _get_varint(handle, 0); bsz, p = _get_varint(handle, p)
        entries += _read_block(buf, boff, bsz)
    return entries


def _block(entries):
    """entries -> block bytes (restart interval 1: every key stored in full) incl. the 5-byte trailer."""
    body, restarts = bytearray(), []
    for k, v in entries:
        restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(k)) + _put_varint(len(v)) + k + v
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body) + b"\x00" + struct.pack("<I", mask_crc(crc32c(bytes(body) + b"\x00")))


def _handle(off, size):
    return _put_varint(off) + _put_varint(size)


# ---- public API ----------------------------------------------------------------------------------------------------
class CheckpointReader(object):
    """Same surface as tf.train.NewCheckpointReader for what weights_utils needs."""

    def __init__(self, prefix, verify_crc=True):
        self.prefix, self.verify = prefix, verify_crc
        if not os.path.exists(prefix + ".index"):
            raise IOError("no TensorFlow V2 checkpoint at %r (missing %s.index; V1 .ckpt files are not supported)" % (prefix, prefix))
        self.entries, self.num_shards = {}, 1
        for k, v in _read_table(prefix + ".index"):
            if k == b"":
                for f, w, val in _parse_proto(v):
                    if f == 1: self.num_shards = val
                    if f == 2 and val != 0: raise NotImplementedError("big-endian checkpoints are not supported")
            else:
                self.entries[k.decode()] = _parse_entry(v)
        self._shards = {}

    def has_tensor(self, name):
        return name in self.entries

    def get_variable_to_shape_map(self):
        return {k: list(e["shape"]) for k, e in self.entries.items()}

    def get_variable_to_dtype_map(self):
        return {k: DT_NUMPY.get(e["dtype"]) for k, e in self.entries.items()}

    def _shard(self, i):
        if i not in self._shards:
            self._shards[i] = np.memmap("%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards), dtype=np.uint8, mode="r")
        return self._shards[i]

    def get_tensor(self, name):
        e = self.entries[name]
        if e["slices"]:
            raise NotImplementedError("partitioned variable %r" % name)
        if e["dtype"] not in DT_NUMPY:
            raise NotImplementedError("dtype enum %d of %r" % (e["dtype"], name))
        raw = self._shard(e["shard_id"])[e["offset"]:e["offset"] + e["size"]]
        if self.verify and e["crc32c"] is not None and mask_crc(crc32c(raw.tobytes())) != e["crc32c"]:
            raise ValueError("checksum mismatch for %r" % name)
        return np.frombuffer(raw.tobytes(), dtype=DT_NUMPY[e["dtype"]]).reshape(e["shape"]).copy()


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: parse the text-proto `checkpoint` state file."""
    f = os.path.join(directory, "checkpoint")
    if not os.path.exists(f):
        return None
    for line in open(f):
        if line.startswith("model_checkpoint_path:"):
            p = line.split(":", 1)[1].strip().strip('"')
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def is_checkpoint(path):
    return os.path.exists(path + ".index")


def write_checkpoint(prefix, tensors, block_entries=64):
    """{name: ndarray} -> <prefix>.index + <prefix>.data-00000-of-00001 (+ `checkpoint` state file): lets adapted
    weights go back to TensorFlow users, and is the fixture generator of the reader tests."""
    names = sorted(tensors)
    data, items = bytearray(), []
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, (lambda v: _put_varint(len(v)) + v)(_field(1, 0, _put_varint(1))))
    items.append((b"", header))
    for n in names:
        a = np.asarray(tensors[n]).copy(order='C')            # (ascontiguousarray would promote 0-d to 1-d)
        if a.dtype not in NUMPY_DT:
            raise NotImplementedError("dtype %s" % a.dtype)
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        dims = b"".join(_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_field(1, 0, _put_varint(int(s))))) for s in a.shape)
        ent = _field(1, 0, _put_varint(NUMPY_DT[a.dtype])) + _field(2, 2, _put_varint(len(dims)) + dims)
        if len(data):
            ent += _field(4, 0, _put_varint(len(data)))
        ent += _field(5, 0, _put_varint(len(raw))) + _field(6, 5, struct.pack("<I", mask_crc(crc32c(raw))))
        items.append((n.encode(), ent))
        data += raw
    out, index = bytearray(), []
    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        blk = _block(chunk)
        index.append((chunk[-1][0], _handle(len(out), len(blk) - 5)))
        out += blk
    meta = _block([])
    moff = len(out); out += meta
    iblk = _block(index)
    ioff = len(out); out += iblk
    foot = _handle(moff, len(meta) - 5) + _handle(ioff, len(iblk) - 5)
    out += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (os.path.basename(prefix), os.path.basename(prefix)))
