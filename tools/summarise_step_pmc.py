"""per-kernel means of the PMC counters collected by tools/pmc_step.sh (+ derived ratios)"""
import csv, glob, sys, collections, re
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:48]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for n, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    calls = max(len(v) for v in c.values())
    rows.append((m.get("SQ_BUSY_CYCLES", 0) * calls, n, calls, m))
rows.sort(reverse=True)
print(f"{'kernel':48s} {'calls':>5s} {'busy_us':>8s} {'VALU/wave':>9s} {'MFMA%':>6s} {'LDS/VALU':>8s} {'wait%':>6s} {'ldsconf%':>8s} {'vmem_rd':>9s}")
for tot, n, calls, m in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    busy_us = m.get("SQ_BUSY_CYCLES", 0) / 32 / 2400.0
    valu, mfma = m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_MFMA", 0)
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(f"{n:48s} {calls:5d} {busy_us:8.1f} {valu:9.0f} {100*mfma/max(valu,1):6.1f} {m.get('SQ_INSTS_LDS',0)/max(valu,1):8.2f} "
          f"{100*m.get('SQ_WAIT_INST_ANY',0)/wc:6.1f} {100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_ACTIVE_INST_LDS',1),1):8.1f} {m.get('SQ_INSTS_VMEM_RD',0):9.0f}")
