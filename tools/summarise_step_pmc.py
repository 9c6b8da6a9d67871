"""Per-kernel SQ counter table of tools/pmc_step.sh: python tools/summarise_step_pmc.py DIR [name substring ...]
For every kernel (mean per launch): duration, matrix-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES over 4 SIMDs x 256 CUs x
kernel cycles), vector instructions per launch and their issue share (x 4 cycles), LDS and wait shares of the wave cycles."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
pats = sys.argv[2:]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("Cijk"):
        return "hipBLASLt " + (re.search(r"MT\d+x\d+x\d+", n) or [""])[0]
    return re.sub(r"\(.*", "", n)[:52]


def mean(v):
    return sum(v) / len(v) if v else 0.0


rows = []
for k, c in cnt.items():
    if pats and not any(p in k for p in pats):
        continue
    us = mean(dur.get(k, []))          # under the counter pass (slower than un-profiled; shares are what matters)
    cyc = mean(c["SQ_BUSY_CYCLES"]) / 32 if c["SQ_BUSY_CYCLES"] else us * 2400   # kernel cycles (SQ_BUSY summed over 32 SEs)
    simd_cycles = cyc * 1024
    rows.append((us * len(dur.get(k, [])), short(k), len(dur.get(k, [])), us,
                 mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / simd_cycles if simd_cycles else 0,
                 mean(c["SQ_INSTS_VALU"]), (mean(c["SQ_INSTS_VALU"]) - mean(c["SQ_INSTS_MFMA"])) * 4 / simd_cycles if simd_cycles else 0,
                 mean(c["SQ_INSTS_MFMA"]), mean(c["SQ_INSTS_LDS"]), mean(c["SQ_LDS_IDX_ACTIVE"]) / (cyc * 256) if cyc else 0,
                 mean(c["SQ_LDS_BANK_CONFLICT"]) / max(mean(c["SQ_LDS_IDX_ACTIVE"]), 1),
                 mean(c["SQ_WAIT_INST_ANY"]) / max(mean(c["SQ_WAVE_CYCLES"]), 1), mean(c["SQ_WAVE_CYCLES"]) * 4 / simd_cycles if simd_cycles else 0))
rows.sort(reverse=True)
print(f"{'kernel':52s} {'n':>4s} {'us':>7s} {'mfma%':>6s} {'valu_i':>9s} {'valu%':>6s} {'mfma_i':>8s} {'lds_i':>8s} {'lds%':>5s} {'bconf':>5s} {'wait%':>5s} {'occ':>5s}")
for _, n, calls, us, mf, vi, vs, mi, li, ls, bc, wt, occ in rows[:70]:
    print(f"{n:52s} {calls:4d} {us:7.1f} {100*mf:6.1f} {vi:9.0f} {100*vs:6.1f} {mi:8.0f} {li:8.0f} {100*ls:5.1f} {bc:5.2f} {100*wt:5.1f} {occ:5.2f}")
