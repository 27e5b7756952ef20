"""Sampler parity: bit-exact against vectors produced by the reference module itself
(tests/golden/make_sampler_golden.py) -- the one part of the path the reference pins."""
import json
import os

import numpy as np
import pytest

from Sampler import sampler_factory

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampler_golden.json")))


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: "%s-%d-%d" % (c["name"], c["blocks"], c["seed"]))
def test_sampler_matches_reference(case):
    logits = np.array(case["logits"])
    dist = np.exp(logits) / np.sum(np.exp(logits), axis=0)
    np.random.seed(case["seed"])
    s = sampler_factory.get_sampler(case["name"], case["blocks"], 3)
    draws = [[int(v) for v in np.asarray(s.sample(dist)).reshape(-1)] for _ in range(6)]
    assert draws == case["draws"]


def test_factory_surface():
    assert set(sampler_factory.AVAILABLE_SAMPLER) == {"FIXED", "RANDOM", "ARGMAX", "SEQUENTIAL", "PROBABILITY"}
    assert sampler_factory.get_sampler("FIXED", 1, [2]).sample(np.ones(5) / 5) == [2]
    with pytest.raises(AssertionError):
        sampler_factory.get_sampler("SAMPLE", 1)
