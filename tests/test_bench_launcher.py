"""`bench.py --gpus N` launcher path without a GPU: bench.py re-launches itself under torch.distributed.run (2 ranks), the
ranks rendezvous over gloo on 127.0.0.1, each runs one emulated step (MADNET_HIP_LIB -> tests/emul library, --device cpu:
plumbing only, never a result), the region time is MAX-reduced over the ranks and rank 0 prints the ONE JSON line with
n_gpus = 2.  A launcher that starts a different world size than --gpus must fail loudly."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    from conftest import _emul_backend
    be = _emul_backend()
    env = dict(os.environ)
    env["MADNET_HIP_LIB"] = be.lib.path
    env["MH_EMUL_THREADS"] = "4"
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return env


def test_bench_self_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--height", "60", "--width", "100",
                        "--steps", "1", "--warmup", "0", "--repeats", "1"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-4 * d["value"]          # (the line carries 6 significant digits)
    assert d["config"]["precision"] == "mixed" and d["timing"]["repeats"] == 1


def test_bench_rejects_world_size_mismatch():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_bench_shared_model_two_ranks_two_pieces():
    """--shared-model over 2 gloo ranks: the step all-reduces [estimators + context + loss] asynchronously and [pyramid] behind it, and the
    line carries the collective's own time and the byte counts of the two pieces."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--height", "60", "--width", "100",
                        "--steps", "1", "--warmup", "0", "--repeats", "1", "--shared-model", "--min-region-seconds", "0"], env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    sm = d["shared_model"]
    assert d["n_gpus"] == 2 and len(sm["pieces_bytes"]) == 2 and sm["collective_ms_alone"] > 0
    assert sm["pieces_bytes"][0] > 2 * sm["pieces_bytes"][1] and sum(sm["pieces_bytes"]) == 4 * (3826088 + 4)


def test_bench_set_overrides_are_applied_and_reported(capsys):
    """bench.py --set: module flags / library hooks applied at once, engine attributes returned for every engine built, and the JSON line
    lists the overrides (an A/B run cannot pass for the default configuration)."""
    import types
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    lib = types.SimpleNamespace(tune_conv_rows=lambda v: calls.append(("conv_rows", v)))
    from madnet_hip import engine as E
    try:
        per, sched = bench.apply_overrides(["engine.FUSE_HEAD=False", "engine.EARLY_WGS=128", "eng.fuse_front=False", "tune.conv_rows=0"], lib, E)
        assert per == {"fuse_front": False} and sched == {"FUSE_HEAD": False, "EARLY_WGS": 128} and calls == [("conv_rows", 0)]
        assert E.Schedule().FUSE_HEAD is True and E.Schedule(**sched).EARLY_WGS == 128            # (the default is untouched: the override lives in the engines built with it)
        bench._emit({"value": 1})
        assert json.loads(capsys.readouterr().out)["overrides"][0] == "engine.FUSE_HEAD=False"
        with pytest.raises(AssertionError):
            bench.apply_overrides(["engine.NO_SUCH_FLAG=1"], lib, E)
        # --model dispnet validates against DispNetSchedule: its own fields pass, MADNet-only fields fail loudly (they used to be silently ignored)
        from madnet_hip import dispnet_engine as DE
        _, ds = bench.apply_overrides(["engine.FLUSH_MIN=3", "engine.EARLY_UPDATE=False"], lib, E, DE.DispNetSchedule)
        assert ds == {"FLUSH_MIN": 3, "EARLY_UPDATE": False} and DE.DispNetSchedule(**ds).FLUSH_MIN == 3
        with pytest.raises(AssertionError):
            bench.apply_overrides(["engine.FUSE_HEAD=False"], lib, E, DE.DispNetSchedule)
        with pytest.raises(SystemExit):
            bench.apply_overrides(["other.x=1"], lib, E)
    finally:
        bench._OVERRIDES[:] = []


def test_bench_line_fits_the_drivers_tail_and_parses(capsys, tmp_path, monkeypatch):
    """The LAST stdout line must stay under bench.LINE_BUDGET bytes whatever the detail record weighs (BENCH_r05.json: a 30.8 KB line was not parsed by the
    driver), must json.loads, and must carry the contract's keys + roofline + cpu_baseline; the complete record goes to the detail file."""
    sys.path.insert(0, ROOT)
    import bench
    fam = [{"kernel": "conv_planes_kernel<fwd,bf16x3> %d" % i, "launches": 18.0, "us_per_step": 281.36289 - i, "frac": 0.0902084338, "note": "x" * 300} for i in range(60)]
    roof = {"kernel": "conv_planes_kernel<fwd,bf16x3>", "bound": "mfma", "achieved": 225.5210846667391, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.09020843386669565,
            "traffic": 404845216.0, "algorithmic_bytes_per_step": 330852864.0, "us_per_step": 281.3628979027271, "selection": "y" * 400, "traffic_source": "z" * 300}
    corr = {"kernel": "k" * 100, "bound": "hbm", "achieved": 5504.04, "peak": 8000.0, "unit": "GB/s", "frac": 0.688, "traffic": 570981376, "traffic_source": "t" * 260,
            "launch_ms": 0.0985, "algorithmic_bytes_per_launch": 542638080.0}
    side = {"metric": "m" * 90, "value": 1162.7772096048839, "ms_per_step": 0.86, "timing": {"ms_per_step_all": [0.86] * 40}, "roofline": dict(roof), "kernel_families": fam,
            "cpu_baseline": {"value": 1.3, "cores": 64}, "epe_vs_oracle": 1.95e-4, "within_tolerance": True, "config": {"workload": "w" * 300}}
    full = {"metric": "adapted stereo pairs/sec (whole node), MADNet full-backprop online adaptation 1242x375", "value": 741.1583452669294, "unit": "pairs/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 1.349239344582767, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": bench.DTYPE_LABEL["mixed"],
            "data": "synthetic", "timing": {"repeats": 5, "ms_per_step_min": 1.34, "ms_per_step_max": 1.35, "ms_per_step_all": [1.35] * 5, "timed_steps_per_repeat": 740, "note": "n" * 200},
            "config": {"workload": "MADNet FULL adaptation step (fwd+SSIM/L1 loss+EPE+bwd+momentum), 1242x375, 1 pair/GPU/step, private model per stream", "precision": "mixed",
                       "launch": "hipGraph replay", "timed_region": "r" * 200, "ops_per_step": 138, "final_loss": 0.0055, "epe_vs_synthetic_gt": 4.13},
            "roofline": roof, "kernel_families": fam, "roofline_fwd": dict(roof), "roofline_corr": corr, "roofline_corr_bwd": dict(corr), "roofline_corr_warp_bwd": dict(corr),
            "roofline_corr_d81_fwd": dict(corr), "roofline_corr_d81_bwd": dict(corr),
            "cpu_baseline": {"value": 0.977295994283161, "unit": "pairs/s", "cores": 64, "kind": "port", "sample": "8 FULL steps of the torch-CPU oracle"},
            "epe_vs_oracle": 0.000195, "epe_tolerance": 1e-3, "within_tolerance": True, "step_surface": {"value": 701.4, "unit": "pairs/s", "ms_per_step": 1.4256, "what": "s" * 300},
            "paths": {"fp32": {"dtype": "d" * 100, "value": 265.9, "ms_per_step": 3.76, "epe_vs_oracle": 1.1e-5, "within_tolerance": True}},
            "drift": {"what": "q" * 200, "step_10": {"epe_vs_fp32_engine": 0.0142, "loss": 0.02}, "step_100": {"epe_vs_fp32_engine": 0.085}},
            "configs": {"mad": dict(side), "dispnet": dict(side), "private4": dict(side), "batched4": dict(side)},
            "box": {"replay_over_launch_sum": 0.94, "healthy": True, "during_replay": {"power_w": {"min": 1.0}}, "note": "b" * 300}}
    assert len(json.dumps(full)) > 5 * bench.LINE_BUDGET
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench._emit(full)
    outp = capsys.readouterr().out
    assert outp.endswith("\n") and outp.count("\n") == 1
    line = outp.strip()
    assert len(line.encode()) < 6144 and len(line.encode()) < bench.LINE_BUDGET
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert abs(d["value"] - full["value"]) <= 1e-5 * full["value"] and d["config"]["workload"] == full["config"]["workload"]
    assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["traffic"] == 404845216
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 64
    assert abs(d["roofline_corr"]["warp_bwd_d5"]["frac"] - 0.688) < 1e-9 and d["configs"]["dispnet"]["roofline"]["frac"] > 0 and d["step_surface"]["value"] == 701.4
    assert d["drift"] == {"step_10": 0.0142, "step_100": 0.085}
    det = json.load(open(os.path.join(str(tmp_path), d["detail"])))
    assert det["kernel_families"][59]["kernel"].endswith(" 59") and det["value"] == full["value"]        # the complete record, full doubles
