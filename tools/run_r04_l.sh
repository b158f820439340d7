#!/bin/bash
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_cabi.py -m gpu -q -x 2>&1 | tail -8
