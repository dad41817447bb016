#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py -q -x 2>&1 | tail -12 | tee $O/tests.log
