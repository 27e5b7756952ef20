"""TensorFlow-checkpoint importer (Data_utils/tf_checkpoint.py, weights_utils.py) -- SURVEY 8(f) rank 1.

No real TF checkpoint can exist in this environment, so the reader is checked against the writer (round trip) and
against hand-assembled bytes of the documented formats (LevelDB table + BundleEntryProto), incl. prefix-compressed
keys and multi-block indices, which the writer itself never produces."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd"))
from Data_utils import tf_checkpoint as CK      # noqa: E402


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert CK.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert CK.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert CK.crc32c(bytes(range(32))) == 0x46DD794E
    assert CK.crc32c(b"123456789") == 0xE3069283
    # leveldb masking is an involution-free rotation + constant
    assert CK.mask_crc(0) == 0xa282ead8
    big = bytes(range(256)) * 1024                         # > 64 KiB: goes through libmadnet_hip's mh_crc32c if present
    t = CK._table(); ref = 0xFFFFFFFF
    for b in big:
        ref = t[(ref ^ b) & 0xFF] ^ (ref >> 8)
    assert CK.crc32c(big) == ref ^ 0xFFFFFFFF
    assert CK.crc32c(big[1000:], CK.crc32c(big[:1000])) == ref ^ 0xFFFFFFFF      # chaining


def test_round_trip_madnet_weights(tmp_path):
    from madnet_hip import engine as E, synthetic as S
    shapes = dict(E.madnet_manifest())
    w = S.xavier_weights(shapes, seed=3)
    w["global_step"] = np.array(1234, dtype=np.int64)
    prefix = str(tmp_path / "model-1234")
    CK.write_checkpoint(prefix, w, block_entries=7)        # many data blocks -> multi-entry index block
    assert CK.is_checkpoint(prefix) and CK.latest_checkpoint(str(tmp_path)) == prefix
    r = CK.CheckpointReader(prefix)
    sm = r.get_variable_to_shape_map()
    assert set(sm) == set(w)
    for k, v in w.items():
        assert sm[k] == list(v.shape)
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and np.array_equal(got, v)
    assert r.get_tensor("global_step") == 1234


def test_reader_handles_prefix_compressed_keys_and_detects_corruption(tmp_path):
    """Hand-built index with shared key prefixes (what TF's table builder emits, restart interval 16)."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.arange(4, dtype=np.float32) * 2
    raw = a.tobytes() + b.tobytes()

    def entry(arr, off):
        dims = b"".join(CK._field(2, 2, (lambda d: CK._put_varint(len(d)) + d)(CK._field(1, 0, CK._put_varint(s)))) for s in arr.shape)
        e = CK._field(1, 0, CK._put_varint(1)) + CK._field(2, 2, CK._put_varint(len(dims)) + dims)
        if off:
            e += CK._field(4, 0, CK._put_varint(off))
        return e + CK._field(5, 0, CK._put_varint(arr.nbytes)) + CK._field(6, 5, struct.pack("<I", CK.mask_crc(CK.crc32c(arr.tobytes()))))

    items = [(b"", CK._field(1, 0, CK._put_varint(1))), (b"model/conv1/biases", entry(b, a.nbytes)), (b"model/conv1/weights", entry(a, 0))]
    body, last = bytearray(), b""
    for i, (k, v) in enumerate(items):
        shared = 0 if i == 0 else len(os.path.commonprefix([last, k]))
        body += CK._put_varint(shared) + CK._put_varint(len(k) - shared) + CK._put_varint(len(v)) + k[shared:] + v
        last = k
    body += struct.pack("<II", 0, 1)                        # one restart point at 0
    blk = bytes(body) + b"\x00" + struct.pack("<I", CK.mask_crc(CK.crc32c(bytes(body) + b"\x00")))
    meta = CK._block([])
    idx = CK._block([(b"model/conv1/weights", CK._handle(0, len(blk) - 5))])
    foot = CK._handle(len(blk), len(meta) - 5) + CK._handle(len(blk) + len(meta), len(idx) - 5)
    table = blk + meta + idx + foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", CK.MAGIC)
    prefix = str(tmp_path / "hand")
    open(prefix + ".index", "wb").write(table)
    open(prefix + ".data-00000-of-00001", "wb").write(raw)
    r = CK.CheckpointReader(prefix)
    assert r.get_variable_to_shape_map() == {"model/conv1/biases": [4], "model/conv1/weights": [2, 3]}
    assert np.array_equal(r.get_tensor("model/conv1/weights"), a) and np.array_equal(r.get_tensor("model/conv1/biases"), b)
    # flip one data byte -> tensor checksum must catch it ; flip an index byte -> block checksum must catch it
    bad = bytearray(raw); bad[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        CK.CheckpointReader(prefix).get_tensor("model/conv1/weights")
    t2 = bytearray(table); t2[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(t2))
    with pytest.raises(ValueError):
        CK.CheckpointReader(prefix)


def test_weights_utils_name_matching(tmp_path):
    """get_var_to_restore_list semantics (weights_utils.py:4-37): prefix, ignore_list, mask."""
    import torch
    from Data_utils import weights_utils as WU
    ck = {"model/a/weights": np.ones((2, 2), np.float32), "model/a/biases": np.full((2,), 3.0, np.float32),
          "model/b/weights": np.zeros((1,), np.float32), "extra/Momentum": np.zeros((1,), np.float32)}
    prefix = str(tmp_path / "ck")
    CK.write_checkpoint(prefix, ck)

    class V(object):
        def __init__(self, shape): self.tensor = torch.full(shape, -1.0)
    net = {"net/a/weights": V((2, 2)), "net/a/biases": V((2,)), "net/b/weights": V((1,))}
    m = WU.get_var_to_restore_list(prefix, mask=["b/"], prefix="net/", ignore_list=["model/"], net=net)
    assert set(m) == {"model/a/weights", "model/a/biases"}
    assert WU.restore(prefix, m) == 2
    assert torch.all(net["net/a/weights"].tensor == 1) and torch.all(net["net/a/biases"].tensor == 3)
    assert torch.all(net["net/b/weights"].tensor == -1)
    ok, step = WU.check_for_weights_or_restore_them(str(tmp_path / "nothing"), net, initial_weights=str(tmp_path), prefix="net/", ignore_list=["model/"])
    assert ok and step == 0 and torch.all(net["net/b/weights"].tensor == 0)


def test_cli_weight_loader_reads_checkpoint(tmp_path):
    import importlib
    from madnet_hip import engine as E, synthetic as S
    soa = importlib.import_module("Stereo_Online_Adaptation")
    w = S.xavier_weights(dict(E.madnet_manifest()), seed=5)
    prefix = str(tmp_path / "weights.ckpt")
    CK.write_checkpoint(prefix, w)
    got = soa.load_weights(prefix, "MADNet")
    assert set(got) == set(w) and all(np.array_equal(got[k], w[k]) for k in w)
    got2 = soa.load_weights(str(tmp_path), "MADNet")        # a directory: latest_checkpoint
    assert np.array_equal(got2["model/gc-read-pyramid/conv1/weights"] if "model/gc-read-pyramid/conv1/weights" in w else got2[sorted(w)[0]],
                          w.get("model/gc-read-pyramid/conv1/weights", w[sorted(w)[0]]))


def test_reader_against_independent_c_writer(tmp_path):
    """VERDICT r01 item 10 / ADVICE: the reader was only ever checked against its own writer.  tests/tools/tb_writer.c is a second,
    independent implementation of the TensorBundle V2 / LevelDB-table format (C, written from the format description, shares no code with
    Data_utils/tf_checkpoint.py): 1 KiB data blocks (several), prefix-compressed keys with restart interval 16, index block with restart
    interval 1, proto3 zero-field omission.  The reader has to recover every tensor bit-exactly and detect a flipped data byte."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "tb_writer")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "tools", "tb_writer.c"), "-lm"], check=True)
    prefix = str(tmp_path / "model-77")
    nt = 40
    subprocess.run([exe, prefix, str(nt)], check=True)
    assert CK.is_checkpoint(prefix)
    r = CK.CheckpointReader(prefix)
    sm = r.get_variable_to_shape_map()
    assert len(sm) == 2 * nt
    for t in range(nt):
        for which, leaf in ((0, "biases"), (1, "weights")):
            shp = (3, 3, t % 5 + 1, t % 7 + 2) if which else (t % 7 + 2,)
            e = np.arange(int(np.prod(shp)), dtype=np.float64)
            ref = np.sin(0.37 * t + 0.011 * e + (0.0 if which else 1.0)).astype(np.float32).reshape(shp)
            name = "model/layer%03d/%s" % (t, leaf)
            assert sm[name] == list(shp)
            got = r.get_tensor(name)
            assert got.dtype == np.float32 and np.array_equal(got, ref), name
    # the index really has several data blocks (else the multi-block path was not exercised)
    assert os.path.getsize(prefix + ".index") > 3 * 1024
    # corruption of the data shard is caught by the per-tensor masked crc32c the C writer computed
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(data, "wb").write(bytes(raw))
    r2 = CK.CheckpointReader(prefix)
    with pytest.raises(Exception):
        for name in sm:
            r2.get_tensor(name)
    # and of the index by the block trailer crc
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[100] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(Exception):
        CK.CheckpointReader(prefix).get_variable_to_shape_map()
