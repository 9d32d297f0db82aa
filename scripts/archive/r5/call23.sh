#!/bin/bash
# round 5, call 23: the whole GPU suite and smoke() on the tree with the matrix-filtered reference-order run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 1100 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=8 ) > gpurun_out/r5/pytest23.log 2>&1; echo "pytest rc=$?"
tail -22 gpurun_out/r5/pytest23.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
