#!/usr/bin/env python3
"""Turn the raw output of tools/refresh_profiles.sh (gpurun_out/<tag>f) into the committed summaries under profiles/.
usage: python tools/make_profiles.py gpurun_out/r02f r02"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02f")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
out = os.path.join(ROOT, "profiles")

line = json.load(open(os.path.join(src, "bench_line.json")))
json.dump(line, open(os.path.join(out, f"{tag}_bench_line.json"), "w"), indent=1)
stats_line = json.load(open(os.path.join(src, "stats_line.json")))

stats = max(glob.glob(os.path.join(src, "stats/*/*kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
fwd_k = ("hashgrid_fwd", "untranspose_feats", "ctx_")
with open(os.path.join(out, f"{tag}_bench_rocprof_summary.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "
            "--psnr-steps 0 --nerf-steps 0 --no-secondary\n\n")
    f.write(f"Workload S1 (3-D nerf_hash grid L16 F2 bw19, N = 2^20 per step), MI355X, {tag} (final kernels, source hash "
            f"{kernel_source_hash()}). 35 calls = 5 warm-up + 30 timed.\n")
    f.write(f"bench.py's own HIP-event timing in the same run: forward {stats_line['ms']['forward']:.3f} ms, backward "
            f"{stats_line['ms']['backward']:.3f} ms per step ({stats_line['value'] / 1e6:.0f} M samples/s).\n\n")
    f.write("| kernel | calls | avg us | % |\n|---|---|---|---|\n")
    tf = tb = 0.0
    for r in rows:
        if int(r["Calls"]) < 30 and "rocclr" not in r["Name"]:
            continue
        avg = float(r["AverageNs"]) / 1e3
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {avg:.1f} | {float(r['Percentage']):.2f} |\n")
        per_step = float(r["TotalDurationNs"]) / 35 / 1e3
        if any(k in r["Name"] for k in fwd_k):
            tf += per_step
        elif "shacira::" in r["Name"] or "fillBuffer" in r["Name"]:
            tb += per_step
    f.write(f"\nforward operator = sample sort (ctx_count + 2 scans + ctx_scatter) + hashgrid_fwd_level_pair (fine levels) + "
            f"hashgrid_fwd_rows (coarse levels + row assembly) = {tf:.1f} us of kernel time; backward "
            f"operator = transpose_grad + bin_count + 2 scans + bin_scatter + bin_consume + direct_accumulate + memset = "
            f"{tb:.1f} us of kernel time per step. The backward's event time is shorter than its kernel-time sum because "
            f"bin_count and the scans run on the library's side stream concurrently with transpose_grad and "
            f"direct_accumulate (DESIGN.md 4.3).\n")

def counter(dirname):
    f = max(glob.glob(os.path.join(src, dirname, "*/*counter_collection.csv")), key=os.path.getmtime)
    acc, calls = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[name] += float(r["Counter_Value"])
        calls[name] += 1
    return acc, calls

ops = {}
for op, key in (("fwd", "forward"), ("bwd", "backward")):
    fe, calls = counter(f"pmc_{op}_FETCH_SIZE")
    wr, _ = counter(f"pmc_{op}_WRITE_SIZE")
    iters = 3
    kern = {}
    rd = wt = 0.0
    for name in fe:
        if "copyBuffer" in name or "at::native" in name or "rocprim" in name:
            continue
        kern[name] = {"FETCH_SIZE_KiB_raw": fe[name] / iters, "WRITE_SIZE_KiB_raw": wr.get(name, 0.0) / iters}
        rd += 2 * fe[name] / iters * 1024
        wt += wr.get(name, 0.0) / iters * 1024
    ops[key] = {"read_bytes_corrected": rd, "write_bytes": wt, "hbm_bytes_per_launch": rd + wt, "kernels": kern}
json.dump({"kernel_source_hash": kernel_source_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python3 tools/fwd_only.py "
                   "{fwd|bwd} -1 3 3` (workload S1: 3-D L16 F2 bw19, N=2^20; 3 calls, per-call averages). Counters are in "
                   "KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section), so read "
                   "bytes = 2*FETCH_SIZE*1024 (verified: the transposes read 128 MiB and report 64 MiB); WRITE_SIZE is "
                   "exact. fillBufferAligned is the hipMemsetAsync of the gradient table.",
           "operators": ops}, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for k, v in ops.items():
    print(k, "HBM bytes per launch %.3f GB (read %.3f, write %.3f)" % (v["hbm_bytes_per_launch"] / 1e9, v["read_bytes_corrected"] / 1e9, v["write_bytes"] / 1e9))

# extra summaries of this round
for name in ("parity_and_timing.txt", "config_c.txt", "tiled_check.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, f"{tag}_{name}"))

# forward unit counters (judge r1 item 3): per kernel, per call
def ctr(dirname):
    fs = glob.glob(os.path.join(src, dirname, "*/*counter_collection.csv"))
    if not fs:
        return {}
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        n[name][r["Counter_Name"]] += 1
    return {k: {c: v / n[k][c] for c, v in d.items()} for k, d in acc.items()}

with open(os.path.join(out, f"{tag}_fwd_counters.md"), "w") as f:
    f.write(f"# Forward kernels on S1 (N = 2^20), unit counters per launch ({tag}, source hash {kernel_source_hash()})\n\n"
            "`rocprofv3 --pmc <group> --kernel-trace -- python3 tools/fwd_only.py fwd -1 3 3`, one pass per counter group "
            "(SQ / TA / TCP / TCC), averages over 3 calls. `tiled` = cell-sorted path (default), `plain` = "
            "SHACIRA_OPTIONS=tiled=0 (level-per-XCD pair kernel over all 16 levels + untranspose).\n")
    for mode in ("tiled", "plain"):
        merged = defaultdict(dict)
        for grp in ("sq", "ta", "tcp", "tcc"):
            for k, d in ctr(f"ctr_{mode}_{grp}").items():
                if "shacira::" in k:
                    merged[k].update(d)
        f.write(f"\n## {mode}\n\n")
        for k, d in merged.items():
            f.write(f"### `{k}`\n\n| counter | per launch |\n|---|---|\n")
            for c in sorted(d):
                f.write(f"| {c} | {d[c]:,.0f} |\n")
            gui = d.get("GRBM_GUI_ACTIVE"); ta = d.get("TA_TA_BUSY_sum"); hit = d.get("TCC_HIT_sum"); miss = d.get("TCC_MISS_sum")
            notes = []
            if gui and ta:
                notes.append(f"TA busy = TA_TA_BUSY_sum / (256 TAs x GRBM_GUI_ACTIVE / 8 XCDs) = {ta / (256 * gui / 8):.2f}")
            if hit is not None and miss is not None and hit + miss > 0:
                notes.append(f"L2 hit rate = {hit / (hit + miss):.3f}")
            if d.get("SQ_WAVE_CYCLES"):
                notes.append(f"waves parked in waits = SQ_WAIT_ANY / SQ_WAVE_CYCLES = {d.get('SQ_WAIT_ANY', 0) / d['SQ_WAVE_CYCLES']:.2f}")
            if d.get("SQ_WAVES") and d.get("SQ_INSTS_VALU"):
                notes.append(f"VALU instructions per wave = {d['SQ_INSTS_VALU'] / d['SQ_WAVES']:.0f}, VMEM reads per wave = {d.get('SQ_INSTS_VMEM_RD', 0) / d['SQ_WAVES']:.1f}")
            f.write("\n" + "; ".join(notes) + "\n\n")
print("wrote", f"{tag}_fwd_counters.md")
