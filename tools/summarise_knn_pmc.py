"""average the PMC counters of a kNN kernel from tools/pmc_knn*.sh output directories: summarise_knn_pmc.py DIR [name substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("/") else "knn_rows_mfma"
dirs = [a for a in sys.argv[1:] if a != pat]
for d in dirs:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and "prep" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for k in sorted(acc):
        v = acc[k]
        print(f"  {k:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
