#!/bin/bash
set -u
timeout 900 python scripts/convergence_check.py 2>&1 | grep -v Warning | tail -8
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-700
