"""BASELINE config 1 plumbing: the online-adaptation CLI over a list in example_list.csv format
(generated frames), modes NONE and MAD, writes stats.csv / series.csv / 16-bit disparity PNGs."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real-time-self-adaptive-deep-stereo_amd")


def _make_list(tmp_path, n, H, W):
    from PIL import Image
    from madnet_hip import synthetic as S
    rows = []
    for t in range(n):
        l, r, gt = S.make_pair(H, W, frame=t)
        names = [str(tmp_path / ("%s_%d.png" % (k, t))) for k in ("l", "r", "d")]
        Image.fromarray(l[0].astype(np.uint8)).save(names[0]); Image.fromarray(r[0].astype(np.uint8)).save(names[1])
        Image.fromarray((gt[0, :, :, 0] * 256).astype(np.uint16)).save(names[2])
        rows.append(",".join(names))
    p = tmp_path / "list.csv"
    p.write_text("\n".join(rows) + "\n")
    return str(p)


@pytest.mark.parametrize("mode", ["NONE", "MAD"])
def test_cli_end_to_end(hip, tmp_path, mode):
    import Stereo_Online_Adaptation as SOA
    lst = _make_list(tmp_path, 3, 96, 160)
    out = tmp_path / ("out_" + mode)
    os.makedirs(out / "disparities")
    argv = ["-l", lst, "-o", str(out), "--weights", "calibrated:1", "--modelName", "MADNet",
            "--blockConfig", os.path.join(PKG, "block_config", "MadNet_full.json"), "--mode", mode,
            "--sampleMode", "SEQUENTIAL", "--imageShape", "96", "160", "--logDispStep", "1", "--SSIMTh", "10"]
    args = SOA.build_parser().parse_args(argv)
    np.random.seed(0)
    SOA.main(args)
    stats = open(out / "stats.csv").read()
    assert stats.startswith("Metrics,cumulative,average\nEPE,") and "FPS," in stats and "#resets,0" in stats
    series = open(out / "series.csv").read().strip().split("\n")
    assert series[0] == "Iteration,Time,EPE,bad3" and len(series) == 4
    from PIL import Image
    d = np.asarray(Image.open(out / "disparities" / "disparity_2.png"))
    assert d.dtype == np.uint16 and d.shape == (96, 160) and d.max() > 0
    if mode == "MAD":
        assert "fetch_counter,1,1,1,0,0" in stats          # SEQUENTIAL sampler over 3 frames
