"""Input list / image reading of the driver (host IO, CPU only): list format of example_list.csv,
centre crop / zero pad semantics, 16-bit KITTI disparity PNG, PFM."""
import os

import numpy as np
import pytest

from Data_utils import data_reader


def test_read_list_file_and_errors(tmp_path):
    p = tmp_path / "list.csv"
    p.write_text("/a/l0.png,/a/r0.png,/a/d0.png\n/a/l1.png,/a/r1.png,/a/d1.png\n")
    l, r, g = data_reader.read_list_file(str(p))
    assert l == ["/a/l0.png", "/a/l1.png"] and r[1] == "/a/r1.png" and g[0] == "/a/d0.png"
    p.write_text("only,two\n")
    with pytest.raises(Exception):
        data_reader.read_list_file(str(p))


def test_center_crop_or_pad_matches_tf_offsets():
    img = np.arange(10 * 12, dtype=np.float32).reshape(10, 12, 1)
    c = data_reader.center_crop_or_pad(img, 6, 8)            # crop offsets (in-target)//2 = (2,2)
    assert c.shape == (6, 8, 1) and c[0, 0, 0] == img[2, 2, 0]
    p = data_reader.center_crop_or_pad(img, 13, 15)          # pad offsets (target-in)//2 = (1,1)
    assert p.shape == (13, 15, 1) and p[1, 1, 0] == img[0, 0, 0] and p[0, 0, 0] == 0 and p[12, 14, 0] == 0


def test_dataset_iterates_png16_and_pfm(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    rows = []
    for i in range(2):
        l = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8); r = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
        d = (rng.random((20, 30)) * 60).astype(np.float32)
        Image.fromarray(l).save(tmp_path / ("l%d.png" % i)); Image.fromarray(r).save(tmp_path / ("r%d.png" % i))
        if i == 0:
            Image.fromarray((d * 256).astype(np.uint16)).save(tmp_path / "d0.png")
            gname = "d0.png"
        else:
            with open(tmp_path / "d1.pfm", "wb") as f:
                f.write(b"Pf\n30 20\n-1.0\n"); np.flipud(d).astype("<f4").tofile(f)
            gname = "d1.pfm"
        rows.append("%s,%s,%s" % (tmp_path / ("l%d.png" % i), tmp_path / ("r%d.png" % i), tmp_path / gname))
        if i == 0:
            d0, l0 = d, l
        else:
            d1 = d
    (tmp_path / "list.csv").write_text("\n".join(rows) + "\n")
    ds = data_reader.dataset(str(tmp_path / "list.csv"), batch_size=1, crop_shape=[16, 32], num_epochs=1)
    assert ds.get_max_steps() == 2
    out = list(ds)
    L, R, G = out[0]
    assert L.shape == (1, 16, 32, 3) and L.dtype == np.float32 and G.shape == (1, 16, 32, 1)
    assert np.array_equal(L[0, :, 1:31], l0[2:18].astype(np.float32))           # crop rows 2.., pad cols by 1
    assert np.allclose(G[0, :, 1:31, 0], np.floor(d0[2:18] * 256) / 256.0)
    assert np.allclose(out[1][2][0, :, 1:31, 0], d1[2:18])


def test_device_prefetcher_preserves_order_and_content(tmp_path):
    """CPU run of the decode-ahead ring (device='cpu': no pinning / streams): every frame arrives once, in order,
    intact, also when the list is longer than the ring."""
    rows = []
    frames = []
    for i in range(7):
        a = np.full((6, 8, 3), float(i), np.float32); g = np.full((6, 8, 1), 10.0 + i, np.float32)
        np.save(tmp_path / ("l%d.npy" % i), a); np.save(tmp_path / ("r%d.npy" % i), a + 0.5); np.save(tmp_path / ("g%d.npy" % i), g)
        rows.append("%s,%s,%s" % (tmp_path / ("l%d.npy" % i), tmp_path / ("r%d.npy" % i), tmp_path / ("g%d.npy" % i)))
        frames.append(i)
    (tmp_path / "list.csv").write_text("\n".join(rows) + "\n")
    ds = data_reader.dataset(str(tmp_path / "list.csv"), batch_size=1, crop_shape=[6, 8], num_epochs=1)
    seen = []
    for l, r, g in data_reader.device_prefetcher(ds, "cpu", depth=2):
        assert l.shape == (1, 6, 8, 3) and g.shape == (1, 6, 8, 1)
        i = int(l[0, 0, 0, 0].item())
        assert float(r[0, 0, 0, 0]) == i + 0.5 and float(g[0, 0, 0, 0]) == 10.0 + i
        seen.append(i)
    assert seen == frames


def test_device_prefetcher_surfaces_reader_errors(tmp_path):
    (tmp_path / "list.csv").write_text("%s,%s,%s\n" % (tmp_path / "missing_l.npy", tmp_path / "missing_r.npy", tmp_path / "missing_g.npy"))
    ds = data_reader.dataset(str(tmp_path / "list.csv"), batch_size=1, crop_shape=[6, 8], num_epochs=1)
    with pytest.raises(Exception):
        for _ in data_reader.device_prefetcher(ds, "cpu"):
            pass
