#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 3; do echo "variant $v"; TEASER_K1_VARIANT=$v python scripts/k1_bench.py 2>&1 | grep -E '"batch": 16|50000'; done
