"""profiles/r1_bench_c2_kernel_stats.csv -> profiles/r1_mfma_utilisation.csv: flop per launch / average kernel duration
against the fp32 matrix peak (157.3 TFLOP/s) for the matrix-core kernels of BASELINE config 2 (B=8, N=2048, k=20)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r1"
rows = list(csv.DictReader([l for l in open(os.path.join(ROOT, "profiles", f"{TAG}_bench_c2_kernel_stats.csv")) if not l.startswith("#")]))
B, N, k = 8, 2048, 20
E = B * N * k
items = [("knn_split_kernel<4,", "kNN graph, 64 channels (coarse sweeps + exact refine; the prep kernel's time is not in this row): "
          "algorithmic 2 B N^2 C flop of the exact fp32 distance block", 2.0 * B * N * N * 64),
         ("ec2_fwd_kernel<64", "EdgeConv layer-2 contraction per edge, forward (2 E 64 64)", 2.0 * E * 64 * 64),
         ("ec2_bwd_kernel<64", "EdgeConv layer-2 backward: y2 recompute + dz1 + dW2 (3 x 2 E 64 64)", 6.0 * E * 64 * 64),
         ("Cijk_Alik_Bljk_S_B_Bias_HA_S_SAV_UserArgs_MT256x256x32", "vendor GEMM, head 192 -> 1024 over 16 384 points, forward",
          2.0 * B * N * 192 * 1024)]
out = ["# MFMA utilisation of the matrix-core kernels of BASELINE config 2 (fp32 MFMA peak 157.3 TFLOP/s, MI355X_MICROARCH.md)",
       f"# flop per launch / rocprofv3 average duration (profiles/{TAG}_bench_c2_kernel_stats.csv); tools/mfma_utilisation.py",
       "kernel,what,flop_per_launch,avg_us,TFLOP_per_s,fraction_of_fp32_matrix_peak"]
for sub, what, fl in items:
    r = next((r for r in rows if sub in r["Name"]), None)
    if r is None:
        continue
    a = float(r["AverageNs"]) / 1e3
    tf = fl / (a * 1e-6) / 1e12
    out.append(f"\"{sub}\",\"{what}\",{fl:.3e},{a:.1f},{tf:.1f},{tf / 157.3:.3f}")
open(os.path.join(ROOT, "profiles", f"{TAG}_mfma_utilisation.csv"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
