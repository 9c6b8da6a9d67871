"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as MI355X_MICROARCH.md prescribes) into
profiles/hbm_traffic.json + profiles/<tag>_pmc_hbm_traffic.csv.

    python tools/summarise_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (FETCH_SIZE counts a 128-byte request as 64 bytes).
The "c2" entry is the forward kNN + neighbour-gather group of one BASELINE-config-2 step (bench.py's roofline group)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, tag = sys.argv[1:4]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    ours = sorted(k for k in f if re.match(r"(knn_|ec1_|ec2_|ec2s_|bn|csr_|sum_partials|nnu_|pt_|pw_|gemm_small|fps_|chamfer|adam_)", k))
    kernels = {}
    for k in ours:
        fk = sum(f[k]) / len(f[k])
        wk = sum(w.get(k, [0.0])) / max(1, len(w.get(k, [0.0])))
        kernels[k] = {"launches": len(f[k]), "fetch_size_kb_raw": round(fk, 1), "write_size_kb": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    group = {}
    for k in kernels:   # forward graph build + neighbour gather: per-step launch counts (round 4 kernel set)
        if k.startswith("knn_nominate_kernel<1, true") or k.startswith("knn_refine_kernel<4>") or k.startswith("knn_split_prep_kernel<1, true"):
            group[k] = 1
        elif k.startswith("knn_nominate_kernel<4,") or k.startswith("knn_refine_kernel<64>"):
            group[k] = 2
        elif k.startswith("ec1_stats_select_kernel<true>"):
            group[k] = 2
        elif k.startswith("ec1_stats_select_kernel<false>") or k.startswith("ec2s_fwd_kernel") or k.startswith("ec1_apply_kernel"):
            group[k] = 1
        elif k.startswith("ec1_apply_prep_kernel"):
            group[k] = 2
        elif k.startswith("bn_finalize_kernel"):
            group[k] = 4
    total = sum(kernels[k]["hbm_bytes_per_launch"] * n for k, n in group.items())
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) over "
                       "`python3 bench.py --steps 5 --warmup 3 --eager --no-cpu-baseline`; HBM bytes per launch = "
                       "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B); produced by "
                       "tools/summarise_pmc.py", "c2": total, "c2_group": group, "kernels": kernels}
    json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.csv"), "w") as fh:
        fh.write("# see hbm_traffic.json/_comment ; averages per launch, BASELINE config 2\n")
        fh.write("kernel,launches,FETCH_SIZE_KB_raw,WRITE_SIZE_KB,hbm_bytes_per_launch\n")
        for k, v in kernels.items():
            fh.write(f"\"{k}\",{v['launches']},{v['fetch_size_kb_raw']},{v['write_size_kb']},{v['hbm_bytes_per_launch']}\n")
    print("group bytes per step:", total, group)


if __name__ == "__main__":
    main()
