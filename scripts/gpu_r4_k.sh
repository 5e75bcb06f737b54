#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_convex_ops.py tests/test_gpu_reference_kernels.py -q -k "convex" 2>&1 | tail -15
