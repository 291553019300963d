#!/usr/bin/env python3
"""legacy_threads.py -- examples/host_legacy_threads (the daemon's threading over the legacy signatures beside a batch) for several
batch sizes: what the heartbeat costs alone and in that company.  One JSON line per batch size."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_legacy_threads_gpu import HOST, _inputs
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1024,16384,65536").split(",")]
    d = tempfile.mkdtemp(prefix="legacy_threads_")
    _inputs(d, n)
    for b in sizes:
        r = subprocess.run([HOST, d, str(n), str(b)], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            sys.exit(1)
        print(r.stdout.strip().splitlines()[-1])
        sys.stdout.flush()


if __name__ == "__main__":
    main()
