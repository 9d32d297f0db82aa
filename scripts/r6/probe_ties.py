import json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np
import fluidaudio_amd as fa
from ahc_full_inputs import ahc_tied_input
ctx = fa.default_context(0)
for kind in ("dup30", "silence5", "grid64"):
    xd = ahc_tied_input(kind)
    print("=====", kind, flush=True); sys.stderr.write("===== %s\n" % kind); sys.stderr.flush()
    st, z, s = fa.linkage(xd, ctx=ctx, return_stats=True)
