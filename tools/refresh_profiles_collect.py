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
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "mfma_utilisation.py")])
print("bench:", bench["ms_per_step"], "ms/step", bench["value"], bench["unit"], "roofline", bench["roofline"]["frac"],
      "knn", bench["roofline_knn"]["frac"], "cpu", bench.get("cpu_baseline", {}).get("value"))
