"""GPU parity: the rolling-window tap loop of the RoIAlign forward (round 6) against the per-bin loop it replaced."""
import pytest

pytestmark = pytest.mark.gpu


def test_rolling_window_equals_the_per_bin_loop():
    """Round 6: the product tap loop of the channels-last merged forward is a rolling window of 4 groups of 4 rows
    (hand-counted vmcnt, csrc/roi_align_impl.inc); the per-bin loop of rounds 1-5 stays behind JDET_ROI_FWD_GRAN=4.  Both
    fold the same entries in the same order with the same fmaf: every output word must be EQUAL -- 12 shapes x dialects
    (north-star size, masked RoIs, RoIs across and beyond the border: empty bins, every channel count class, 1x1 .. 8x3
    grids, RiRoIAlign with 4 / 8 orientations).  The switch is read once per process: scripts/r6_ring_check.py runs
    itself once per setting and compares the files."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r6_ring_check.py"), "4", "256"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL BIT-EQUAL" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
