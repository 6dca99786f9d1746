#!/usr/bin/env python3
"""pick.py key[,key...] : from JSON lines on stdin print the named keys of each line (helper of scripts/gpu.sh)."""
import json
import sys

keys = sys.argv[1].split(",")
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("  " + "  ".join("%s=%s" % (k, ("%.4g" % d[k]) if isinstance(d.get(k), float) else d.get(k)) for k in keys))
