#!/bin/bash
python -m pytest tests/test_gpu_refapp.py tests/test_gpu_ofdmtxrx.py -q -x 2>&1 | tail -8
