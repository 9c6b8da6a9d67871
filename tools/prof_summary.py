"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv: one line per kernel, short names, per-step microseconds.
usage: python tools/prof_summary.py <kernel_stats.csv> <steps> [top]"""
import csv, re, sys
rows = list(csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#")))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("Cijk"):
        m = re.search(r"MT(\d+x\d+x\d+)", n); return "hipBLASLt " + (n[:14]) + " MT" + (m.group(1) if m else "")
    n = re.sub(r"\(.*", "", n)
    return n[:90]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.1f us/step over %d kernels" % (tot / steps / 1e3, len(rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print("%8.1f us/step  %6.1f calls/step  avg %8.2f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e3, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, short(r["Name"])))
