"""gpurun_out/refresh/ (tools/refresh_profiles.sh) -> profiles/: kernel statistics with a header, the bench lines, the PMC
HBM-traffic tables and the MFMA utilisation table."""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "refresh")
prof = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
def newest(pattern):     # gpurun merges new files into gpurun_out/: older runs may still be lying around
    return max(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)


stats = newest(os.path.join("stats", "**", "*kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "stats.log")) if '"ms_per_step"' in l][-1]
prof_ms = json.loads(line)["ms_per_step"]
bench = json.loads([l for l in open(os.path.join(src, "bench_c2.json")) if l.startswith("{")][-1])
with open(os.path.join(prof, f"{tag}_bench_c2_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline\n")
    f.write("# BASELINE config 2 (DGCNN-seg N=2048 k=20, 8 clouds, fp32, CE + generalised Dice, FlatAdam), training step replayed as a\n")
    f.write("# hipGraph; 100 timed replays + warm-up/capture steps + 5 eager steps for the per-entry-point HIP-event timing + 25 replays\n")
    f.write(f"# of the forward kNN+gather group; {prof_ms} ms/step under the profiler, {bench['ms_per_step']} ms/step un-profiled\n")
    f.write(open(stats).read())
json.dump(bench, open(os.path.join(prof, f"{tag}_bench_c2.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench_other.jsonl"), os.path.join(prof, f"{tag}_bench_other_configs.jsonl"))
fetch = newest(os.path.join("pmc_fetch", "**", "*counter_collection.csv"))
write = newest(os.path.join("pmc_write", "**", "*counter_collection.csv"))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarise_pmc.py"), fetch, write, tag])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "mfma_utilisation.py"), tag])
# kernel statistics of the other configurations (configs 3, 4, 5) and the SQ counter summary of the dominant kernel
for w in ("c3", "c4", "c5"):
    try:
        st = newest(os.path.join(f"stats_{w}", "**", "*kernel_stats.csv"))
    except ValueError:
        continue
    ln = [l for l in open(os.path.join(src, f"stats_{w}.log")) if '"ms_per_step"' in l][-1]
    with open(os.path.join(prof, f"{tag}_bench_{w}_kernel_stats.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload {w} --steps 50 --warmup 5 --no-cpu-baseline --min-seconds 0.1\n")
        f.write(f"# {json.loads(ln)['config']['workload']}; {json.loads(ln)['ms_per_step']} ms/step under the profiler (warm-up, capture and the eager\n")
        f.write("# per-entry-point timing steps are part of the trace: divide Calls by the fsg adam_flat_kernel row for per-step counts)\n")
        f.write(open(st).read())
sq = os.path.join(src, "knn_sq_counters.txt")
if os.path.exists(sq):
    with open(os.path.join(prof, f"{tag}_knn_sq_counters.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python3 tools/knn_split_prof.py 8 64 2048 20 (fsg_knn_dense_ws_f32, B=8 N=2048 C=64 k=20);\n")
        f.write("# means over the launches of knn_nominate_kernel<4,false,true,true,false> and knn_refine_kernel<64> (two blocks), summed over the device's SEs/XCDs as rocprofv3 reports them.\n")
        f.write("# SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) = matrix-pipe busy cycles per SIMD; SQ_BUSY_CYCLES / 32 = kernel cycles.\n")
        f.write(open(sq).read())
for nm in ("knn_phase_stamps.txt", "knn_split_timing.txt", "step_sq_counters.txt"):
    if os.path.exists(os.path.join(src, nm)):
        shutil.copy(os.path.join(src, nm), os.path.join(prof, f"{tag}_{nm}"))
rl = bench["roofline"]
print("bench:", bench["ms_per_step"], "ms/step", bench["value"], bench["unit"], "roofline", rl["bound"], rl["frac"],
      "group hbm", bench["roofline_group_hbm"]["frac_vs_reference_bytes"], "cpu", bench.get("cpu_baseline", {}).get("value"))
