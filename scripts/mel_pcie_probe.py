#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer mel entry (fa_mel_batch: H2D copy, kernel, D2H copy, sync) on BASELINE configs[1]."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402

B, N = 1024, 240000
ctx = fa.default_context()
L = fa._lib
cfg = L.MelConfig()
fa.lib().fa_mel_default_config(C.byref(cfg))
rng = np.random.default_rng(0)
pcm = (0.1 * rng.standard_normal(B * N)).astype(np.float32)
offs = (np.arange(B + 1, dtype=np.int64) * N)
T = fa.lib().fa_mel_num_frames(C.byref(cfg), N)
mel = np.zeros((B, 128, T), np.float32)
lens = np.zeros(B, np.int32)


def run():
    ctx.check(fa.lib().fa_mel_batch(ctx.handle, C.byref(cfg), pcm.ctypes.data, offs.ctypes.data, B, None, None, T, mel.ctypes.data, lens.ctypes.data), "fa_mel_batch")


run()
t = []
for _ in range(3):
    t0 = time.perf_counter(); run(); t.append(time.perf_counter() - t0)
best = min(t)
print(json.dumps({"mel_host_pointer_entry": {"chunks": B, "seconds": best, "audio_hours_per_s": B * 15 / 3600 / best,
                                             "bytes_moved": int(pcm.nbytes + mel.nbytes), "GBps_over_pcie": (pcm.nbytes + mel.nbytes) / best / 1e9,
                                             "note": "pageable host memory, synchronous copy-in / kernel / copy-out"}}))
