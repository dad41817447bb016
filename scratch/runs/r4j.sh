#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
VARIANTS="new cadence" bash scratch/r4g.sh
