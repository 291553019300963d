"""Random shapes of the record heartbeat through wmx_chain_process against per-handle oracle runs (tools_dev/fuzz_parity.py: channels,
rate, interval, stage switches incl. the fixed-point builds, batch size, packets per call, four buffer layouts with and without
padding, in place / out of place, AGC gain, reported delay).  A short seeded campaign on every GPU run; the long ones are under
profiles/r05/fuzz_parity_summary.txt (4 500 cases, 0 failures)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_heartbeat_shapes_equal_the_oracle(cuda, oracle_port):
    sys.path.insert(0, os.path.join(ROOT, "tools_dev"))
    import fuzz_parity as F
    rng = np.random.default_rng(20251)
    seen = set()
    for i in range(60):
        c = F.draw(rng)
        bad, pad_ok = F.run_case(c, cuda, oracle_port, 81000 + 53 * i)
        assert bad == 0 and pad_ok, c
        assert c["samples_the_chain_changed"] > 0, c
        seen.add((c["layout"], c["interval_ms"], c["chn"]))
    assert len(seen) >= 9  # the campaign is spread over layouts, cadences and channel counts (12 combinations exist)
