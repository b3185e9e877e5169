"""Generates tests/golden/ref_chain_pairs.npz from the COMPILED REFERENCE (oracle/_ref/libtsd_ref.so,
built from /root/reference's PairAssignment.cpp / DistanceFilter.cpp / ReciprocalFilter.cpp).
Run in the build container:  python tests/golden/make_ref_chain_fixture.py
The fixture holds inputs and the reference's outputs only (pair lists per determinePairs call)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from tests.test_cpu_oracle_ref import clouds, ref_chain_run  # noqa: E402

O.build(with_ref=True)
out = {}
cases = [(30, 101, 300, 350), (25, 102, 200, 260), (11, 103, 150, 150), (10, 104, 150, 180), (4, 105, 120, 100)]
for c, (iters, seed, nm, ns) in enumerate(cases):
    model, scene = clouds(seed, nm, ns)
    bounds = (4.2, 8.4, 0.0, 100.0)
    premask = ~((scene[:, 0] < bounds[0]) | (scene[:, 0] > bounds[1]) | (scene[:, 1] < bounds[2]) | (scene[:, 1] > bounds[3]))
    calls = 12
    ref = ref_chain_run(model, scene, premask, iters, 0.4, 0.02, calls)
    out[f"model_{c}"], out[f"scene_{c}"] = model, scene
    out[f"iters_{c}"], out[f"calls_{c}"], out[f"bounds_{c}"] = iters, calls, np.array(bounds)
    for k, (pm, ps) in enumerate(ref):
        out[f"pm_{c}_{k}"], out[f"ps_{c}_{k}"] = pm, ps
out["n_cases"] = len(cases)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_chain_pairs.npz"), **out)
print("wrote", len(cases), "cases")
