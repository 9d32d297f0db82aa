#!/bin/bash
# round 6, call 7: AUTO's tie route hands back to the rounds once the ties have stopped: parity tests + the tied 43 200 x 256 inputs timed
python -m pytest tests/test_gpu_ahc_handover.py tests/test_gpu_ahc_tied_digest.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_ahc.py -x -q -p no:cacheprovider 2>&1 | tail -n 12
python - <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np
import fluidaudio_amd as fa
from ahc_full_inputs import ahc_tied_input, dendrogram_digest
ctx = fa.default_context(0)
x = ahc_tied_input("tie_free")
for _ in range(2):
    t0 = time.perf_counter(); st, z, s = fa.linkage(x, ctx=ctx, return_stats=True); tf = time.perf_counter() - t0
print(json.dumps({"kind": "tie_free", "seconds": tf, "stats": s}))
for kind in ("dup30", "silence5", "grid64"):
    want = json.load(open(f"tests/golden/ahc_tied_{kind}_43200.json"))
    xd = ahc_tied_input(kind)
    for _ in range(2):
        t0 = time.perf_counter(); st, z, s = fa.linkage(xd, ctx=ctx, return_stats=True); t = time.perf_counter() - t0
    print(json.dumps({"kind": kind, "seconds": t, "over_tie_free": t / tf, "equals_reference_digest": dendrogram_digest(z)["dendrogram_sha256"] == want["dendrogram_sha256"], "stats": s}))
PY
