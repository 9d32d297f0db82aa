"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/, itself pinned against the
reference's fixtures and the reference's own C++ build — see tests/test_oracle_*.py).

The reference is Swift + Accelerate and cannot run here, so these vectors are outputs of the
restatement (and, for AHC, of the reference's own FastClusterWrapper C++ built by oracle/Makefile).
They exist so that the GPU parity tests have committed expected values that do not depend on
the oracle build being present.      python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from conftest import speaker_mixture, synth_audio  # noqa: E402


def main():
    oracle.build()
    # mel: 1.0 s and a ragged 0.7731 s utterance, both layouts
    a1, a2 = synth_audio(16000, 11), synth_audio(12370, 12)
    m1, l1, n1 = oracle.mel_flat(a1)
    m2, l2, n2 = oracle.mel_flat_transposed(a2, last=0.25)
    m3, l3, n3 = oracle.mel_flat_transposed(a2, prepadded=True)
    np.savez_compressed(os.path.join(HERE, "mel_golden.npz"), a1=a1, a2=a2, flat1=m1, len1=l1, tr2=m2, len2=l2,
                        pre3=m3, len3=l3)
    # ctc: 3 matrices T=200 V=65 (blank 64, biased) fp32
    rng = np.random.default_rng(7)
    lg = rng.standard_normal((3, 200, 65)).astype(np.float32)
    lg[:, :, 64] += 2.0
    ids = [oracle.ctc_greedy(lg[b], 64) for b in range(3)]
    fids = np.stack([oracle.argmax_rows(lg[b]) for b in range(3)])
    np.savez_compressed(os.path.join(HERE, "ctc_golden.npz"), logits=lg, frame_ids=fids,
                        lens=np.array([len(i) for i in ids]), tokens=np.concatenate(ids))
    # ahc: 600 x 64 mixture + 300 x 32 iid, dendrograms from the reference's own C++ (oracle/_ref)
    xm = speaker_mixture(600, 64, 12, 0.05, 3)
    xi = oracle.ahc_normalize(np.random.default_rng(5).standard_normal((300, 32)))
    sm, zm = oracle.linkage_ref(xm)
    si, zi = oracle.linkage_ref(xi)
    assert sm == 0 and si == 0
    np.savez_compressed(os.path.join(HERE, "ahc_golden.npz"), xm=xm, zm=zm, xi=xi, zi=zi,
                        labels_m=oracle.ahc_cut(zm, 600, 0.6), labels_i=oracle.ahc_cut(zi, 300, 1.2))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
