#!/usr/bin/env python
"""Re-render INTEGRATION.md section 4's table from krypy_amd/_switches.py (tests/test_abi.py holds the two to each other)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krypy_amd import _switches  # noqa: E402

path = os.path.join(ROOT, "INTEGRATION.md")
lines = open(path).read().split("\n")
i0 = next(i for i, ln in enumerate(lines) if ln.startswith("| variable | default | kind | meaning |"))
i1 = i0
while i1 < len(lines) and lines[i1].startswith("|"):
    i1 += 1
lines[i0:i1] = _switches.markdown().split("\n")
open(path, "w").write("\n".join(lines))
print("INTEGRATION.md: %d rows" % len(_switches.names()))
