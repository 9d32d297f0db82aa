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


out = {}
pageable = (pcm, mel)
p_pcm, p_mel = L.pinned_array(pcm.shape, np.float32), L.pinned_array(mel.shape, np.float32)
p_pcm[:] = pcm
for name, mb, pin in (("pageable", None, False), ("pinned_one_slice", "0", True), ("pinned_pipelined_64MB_slices", None, True),
                      ("pinned_pipelined_32MB_slices", "32", True), ("pinned_pipelined_128MB_slices", "128", True)):
    pcm, mel = (p_pcm, p_mel) if pin else pageable
    if mb is None:
        os.environ.pop("FA_MEL_SLICE_MB", None)
    else:
        os.environ["FA_MEL_SLICE_MB"] = mb
    run()
    t = []
    for _ in range(4):
        t0 = time.perf_counter(); run(); t.append(time.perf_counter() - t0)
    best = min(t)
    out[name] = {"seconds": best, "audio_hours_per_s": B * 15 / 3600 / best, "GBps_both_directions": (pcm.nbytes + mel.nbytes) / best / 1e9,
                 "GBps_up": pcm.nbytes / best / 1e9, "GBps_down": mel.nbytes / best / 1e9}
print(json.dumps({"mel_host_pointer_entry": {"chunks": B, "bytes_up": int(pcm.nbytes), "bytes_down": int(mel.nbytes), "variants": out,
                                             "note": "pageable = ordinary host memory (staged copies, one slice); pinned = fa_host_alloc buffers; one_slice = copy-in, kernel, copy-out in "
                                                     "sequence; pipelined = upload of slice k+1 overlaps the download of slice k on a second stream; PCIe Gen5 x16 = 63 GB/s per direction"}}))
