#!/bin/bash
# (ran on an intermediate tree whose Gram kernel took FA_AHC_GRAM_STAGGER; the switch and the code left the tree with the result: profiles/r06_gram_minima.txt)
# round 6, call 2: does a staggered start of the two workgroups of a CU hide the Gram epilogue under the other's matrix-core loop?
export FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_ab.so
for s in 0 2 4 6 8 12 16; do
  echo "== stagger $s"
  FA_AHC_GRAM_STAGGER=$s python scripts/ahc_probe.py 43200,43200 --kinds mix --check 0 2>&1 | grep -o '"init_ms": [0-9.]*'
done
