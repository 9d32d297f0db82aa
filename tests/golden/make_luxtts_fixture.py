"""Packs the reference's only mel golden vector into tests/golden/luxtts_prompt.npz so that the DEVICE can be gated on it
on the GPU box (where /root/reference does not exist).

Source (test DATA of the reference, not code): Tests/FluidAudioTests/TTS/LuxTts/Resources/prompt_24k_f32le.bin
(103 936 samples, 24 kHz) and prompt_mel_f32le.bin (406 x 100 float32 = LuxTtsMelExtractor.extract(audio) * featScale 0.1),
used by Tests/FluidAudioTests/TTS/LuxTts/LuxTtsMelExtractorTests.swift:18-41 with a max-abs gate of 1e-3.
    python tests/golden/make_luxtts_fixture.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LUX = "/root/reference/Tests/FluidAudioTests/TTS/LuxTts/Resources"

audio = np.fromfile(os.path.join(LUX, "prompt_24k_f32le.bin"), np.float32)
gold = np.fromfile(os.path.join(LUX, "prompt_mel_f32le.bin"), np.float32).reshape(-1, 100)
assert audio.size == 103936 and gold.shape == (406, 100)
np.savez_compressed(os.path.join(HERE, "luxtts_prompt.npz"), audio=audio, mel_scaled=gold, feat_scale=np.float32(0.1))
print("wrote luxtts_prompt.npz", audio.shape, gold.shape)
