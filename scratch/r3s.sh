#!/bin/bash
python -m pytest tests/test_gpu_stream.py -m gpu -q -x -k "lean_payload or qam_workers or worker_builds" 2>&1 | tail -15
