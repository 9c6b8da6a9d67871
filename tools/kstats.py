"""print name / calls / average ns of a rocprofv3 --stats output directory: python3 tools/kstats.py DIR [substring]"""
import csv
import glob
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.1f} us")
