#!/bin/bash
# round 4: re-validation after the forward mode 3 commit: smoke, full GPU suite, default bench line
set -u
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 900 python bench.py 2>/dev/null | tail -1 | cut -c1-400
