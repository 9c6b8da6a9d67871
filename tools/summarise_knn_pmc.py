"""average the PMC counters of the knn_rows kernel from tools/pmc_knn.sh output directories"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_rows_mfma" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for k in sorted(acc):
        v = acc[k]
        print(f"  {k:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
