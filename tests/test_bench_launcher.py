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
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
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
        with pytest.raises(SystemExit):
            bench.apply_overrides(["other.x=1"], lib, E)
    finally:
        bench._OVERRIDES[:] = []
