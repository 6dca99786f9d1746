#!/usr/bin/env python3
"""energy_summary.py <dir> <lib> ...: mean power / clock of the steady part of each run (the samples above 90 % of the run's
median power) x the step time -> J per step.  Input: power_<lib>_<mode>.txt (clock_watch.sh), run_<lib>_<mode>.jsonl (k1_probe)."""
import json, re, statistics, sys
d = sys.argv[1]
print("%-10s %-5s %8s %8s %8s %8s %8s" % ("lib", "mode", "step_ms", "k1_ms", "power_W", "sclk_MHz", "J/step"))
for lib in sys.argv[2:]:
    for mode in ("one", "pipe"):
        try:
            runs = [json.loads(l) for l in open("%s/run_%s_%s.jsonl" % (d, lib, mode)) if l.startswith("{")]
            nodes, pci = {}, {}
            for l in open("%s/power_%s_%s.txt" % (d, lib, mode)):
                m = re.match(r"# pci (\S+) node (\S+)", l)
                if m:
                    pci[m.group(1).lower()] = m.group(2)
                t = l.split()
                if len(t) >= 6 and t[1] == "sclk" and not l.startswith("#"):
                    # <time> sclk <Hz> mclk <Hz> power [<average uW>] [<input uW>] node <path>
                    pw = [int(x) for x in t[t.index("power") + 1: t.index("node")]] if "node" in t else []
                    nodes.setdefault(t[-1], []).append((int(t[2]) / 1e6, (pw[-1] if pw else 0) / 1e6))
            # our card = the node whose power rises most from the run's first samples (the probe builds its inputs on the
            # host for a second) to its middle; other tenants' cards on the box are steady
            def rise(r):
                mid = r[len(r) // 4: 3 * len(r) // 4] or r
                return statistics.median(x[1] for x in mid) - statistics.mean(x[1] for x in r[:5])
            rows = max(nodes.values(), key=rise) if nodes else []
            mine = [r["pci"].lower() for r in runs if r.get("probe") == "device"]
            if mine and mine[0] in pci:
                rows = nodes[pci[mine[0]]]
            runs = [r for r in runs if r.get("probe") != "device"]
        except OSError:
            continue
        if not runs or not rows:
            continue
        med = statistics.median(r[1] for r in rows)
        st = [r for r in rows if r[1] >= 0.9 * med]
        pw, ck = statistics.mean(r[1] for r in st), statistics.mean(r[0] for r in st)
        for r in runs:
            step = r.get("step_ms_sync", r.get("step_ms"))
            print("%-10s %-5s %8.4f %8.4f %8.0f %8.0f %8.3f%s" % (lib, mode, step, r["k1_ms"], pw, ck, pw * step * 1e-3,
                                                             "  depth %d" % r["depth"] if "depth" in r else ""))
