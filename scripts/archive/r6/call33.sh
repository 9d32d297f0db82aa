#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( time python -m pytest tests/test_gpu_ahc.py -q -p no:cacheprovider -k "register_path_boundary" ) 2>&1 | tail -n 12 | cut -c1-300
