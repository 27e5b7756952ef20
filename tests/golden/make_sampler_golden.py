"""Generates tests/golden/sampler_golden.json by importing the REFERENCE sampler module
(/root/reference/Sampler/sampler_factory.py, numpy-only) in the build container.  Committed so the
vectors travel to boxes where /root/reference does not exist."""
import importlib.util
import json
import sys

import numpy as np

sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_sampler", "/root/reference/Sampler/sampler_factory.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

cases = []
rs = np.random.RandomState(123)
for name in ["RANDOM", "ARGMAX", "SEQUENTIAL", "PROBABILITY", "FIXED"]:
    for nblocks in (1, 2):
        for trial in range(3):
            logits = rs.randn(5) * (0.0 if trial == 0 else 0.5)
            dist = np.exp(logits) / np.sum(np.exp(logits), axis=0)     # Stereo_Online_Adaptation.py:25-27
            seed = 1000 + trial
            np.random.seed(seed)
            s = ref.get_sampler(name, nblocks, 3)
            draws = [[int(v) for v in np.asarray(s.sample(dist)).reshape(-1)] for _ in range(6)]
            cases.append({"name": name, "blocks": nblocks, "seed": seed, "logits": logits.tolist(), "draws": draws})
json.dump({"numpy": np.__version__, "cases": cases}, open("tests/golden/sampler_golden.json", "w"), indent=0)
print(len(cases), "cases")
