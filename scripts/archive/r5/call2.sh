#!/bin/bash
# round 5, call 2: slots per thread of the round kernel — parity of every form with the reference build, round time per CPT
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 1500 python scripts/cpt_probe.py ) > gpurun_out/r5/cpt_probe.log 2>&1; echo "probe rc=$?"
grep -c FAIL gpurun_out/r5/cpt_probe.log
grep -v '"short recording"' gpurun_out/r5/cpt_probe.log | cut -c1-330 | tail -70
grep '"short recording"' gpurun_out/r5/cpt_probe.log | cut -c1-200
