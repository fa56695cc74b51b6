"""Copy the judged evidence of a profiling round from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py r1

Takes the newest rocprofv3 outputs under gpurun_out/prof_<tag>/ (tools/profile_round.sh), keeps the rows
of the dominant kernel from the counter passes, and rewrites profiles/<tag>_traffic.json from the
FETCH_SIZE / WRITE_SIZE passes (gfx950 correction: FETCH_SIZE x 2, MI355X_MICROARCH.md).
"""
import csv
import re
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
KERNEL = "k_chol_dag"
# round 4: the dominant kernel is the RESIDENT launch, k_chol_dag<2, false, false, true> (STREAM); bench.py --steps 5
# --warmup 5 gives it 160 evaluations per dispatch
EVALS_PER_DISPATCH = 160
MODE = "stream"


def is_stream(name):
    head = name.split("(")[0].replace(" ", "")
    # template arguments <C, AUG, LAT, STREAM, WPE>: the fourth one (rounds 1-3 had three, the start of round 4 four)
    m = re.search(r"k_chol_dag<([^>]*)>", head)
    return bool(m) and len(m.group(1).split(",")) >= 4 and m.group(1).split(",")[3] == "true"


def newest(pattern):
    files = glob.glob(os.path.join(src, pattern))
    if not files:
        sys.exit(f"nothing matches {pattern} under {src}")
    return max(files, key=os.path.getmtime)


shutil.copy(newest("trace/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if glob.glob(os.path.join(src, "trace_full/*/*_kernel_stats.csv")):
    shutil.copy(newest("trace_full/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_full_bench.csv"))
if glob.glob(os.path.join(src, "trace_driver/*/*_kernel_stats.csv")):
    # (round 5) the driver's exact command under the kernel trace: stats + every dispatch of the persistent kernels
    shutil.copy(newest("trace_driver/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_driver_command.csv"))
    drows = list(csv.DictReader(open(newest("trace_driver/*/*_kernel_trace.csv"))))
    with open(os.path.join(dst, f"{tag}_kernel_trace_driver_command_dag.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Duration_ms"])
        for r in drows:
            if KERNEL in r["Kernel_Name"]:
                w.writerow([r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Start_Timestamp"], r["End_Timestamp"],
                            "%.4f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)])
    if os.path.exists(os.path.join(src, "trace_driver.log")):
        for ln in open(os.path.join(src, "trace_driver.log")):
            if ln.startswith("{") and '"metric"' in ln:
                open(os.path.join(dst, f"{tag}_bench_under_kernel_trace.json"), "w").write(ln)
shutil.copy(os.path.join(src, f"summary_{tag}.md"), os.path.join(dst, f"{tag}_summary.md"))
# every dispatch of the persistent kernels (resident and launch-per-step) from the kernel trace: the per-dispatch durations
# the roofline of bench.py has to agree with
trace_rows = list(csv.DictReader(open(newest("trace/*/*_kernel_trace.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_trace_dag.csv"), "w", newline="") as f:
    keep_cols = ["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size", "Grid_Size", "VGPR_Count",
                 "Scratch_Size", "LDS_Block_Size"]
    cols = [c_ for c_ in keep_cols if c_ in trace_rows[0]]
    w = csv.DictWriter(f, fieldnames=cols + ["Duration_ms"])
    w.writeheader()
    for r in trace_rows:
        if KERNEL in r["Kernel_Name"]:
            row = {c_: (r[c_].split("(")[0] if c_ == "Kernel_Name" else r[c_]) for c_ in cols}
            row["Duration_ms"] = "%.4f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            w.writerow(row)
means = {}
for name in ("sq", "fetch", "write"):
    rows = list(csv.DictReader(open(newest(f"pmc_{name}/*/*_counter_collection.csv"))))
    keep = [r for r in rows if KERNEL in r["Kernel_Name"]]
    stream_rows = [r for r in keep if is_stream(r["Kernel_Name"])]
    dominant = stream_rows if stream_rows else keep
    if not stream_rows:
        # round 6: the headline is one launch of k_chol_dag<2, false, false, false, 2> per step, 32 evaluations per dispatch
        MODE, EVALS_PER_DISPATCH = "dag", 32
    with open(os.path.join(dst, f"{tag}_pmc_{name}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    for r in dominant:
        means.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
fetch_kb = sum(means["FETCH_SIZE"]) / len(means["FETCH_SIZE"])
write_kb = sum(means["WRITE_SIZE"]) / len(means["WRITE_SIZE"])
tpath = os.path.join(dst, f"{tag}_traffic.json")
old = json.load(open(tpath)) if os.path.exists(tpath) and json.load(open(tpath)).get("workload", {}).get("mode") == MODE else {
    "round": int(tag[1:]) if tag[1:].isdigit() else tag,
    "source": f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, "
              "bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-extras --no-strong)",
    "kernel": ("k_chol_dag<2, false, false, true, 2> (the resident launch of bench.py --mode stream)" if MODE == "stream" else
               "k_chol_dag<2, false, false, false, 2> (one launch per step: bench.py's default --mode dag)"),
    "workload": {"N": 6000, "components": 2, "walkers": 32, "mode": MODE, "evaluations_per_launch": EVALS_PER_DISPATCH},
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16 B/lane coalesced reads); "
            "WRITE_SIZE taken as is"}
old.update(fetch_size_kb_raw=fetch_kb, write_size_kb_raw=write_kb, fetch_correction=2.0,
           hbm_bytes_per_launch=(2.0 * fetch_kb + write_kb) * 1024.0,
           hbm_bytes_per_evaluation=(2.0 * fetch_kb + write_kb) * 1024.0 / EVALS_PER_DISPATCH)
json.dump(old, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=2)
for a, b in ((f"latency_{tag}.jsonl", f"{tag}_latency_table.jsonl"), (f"bench_{tag}.json", f"{tag}_bench.json"),
             (f"gputests_{tag}.txt", f"{tag}_gputests.txt"), (f"soak_{tag}.txt", f"{tag}_soak.txt"),
             (f"stream_table_{tag}.jsonl", f"{tag}_stream_table.jsonl"), (f"timeline_{tag}.txt", f"{tag}_timeline.txt"),
             (f"timeline_launch_per_step_{tag}.txt", f"{tag}_timeline_launch_per_step.txt"), (f"fill_{tag}.jsonl", f"{tag}_fill_table.jsonl"),
             (f"follow_table_{tag}.txt", f"{tag}_follow_table.txt"), (f"row_periods_{tag}.txt", f"{tag}_row_periods.txt"),
             (f"wg_occupancy_{tag}.txt", f"{tag}_wg_occupancy.txt"), (f"lib_sha256_{tag}.txt", f"{tag}_lib_sha256.txt"),
             (f"scheme_table_{tag}.txt", f"{tag}_scheme_table.txt"), (f"queue_sweep_final_{tag}.txt", f"{tag}_queue_sweep_final.txt"),
             (f"shared_gpu_probe_{tag}.txt", f"{tag}_shared_gpu_probe.txt"), (f"evidence_status_{tag}.txt", f"{tag}_evidence_status.txt"),
             (f"predict_timeline_{tag}.txt", f"{tag}_predict_timeline.txt"), (f"part_wait_share_{tag}.txt", f"{tag}_part_wait_share.txt"),
             (f"sampler_stream_{tag}.txt", f"{tag}_sampler_stream_same_call.txt"),
             (f"gather_beside_stream_{tag}.txt", f"{tag}_gather_beside_stream_same_call.txt"),
             (f"small_{tag}.jsonl", f"{tag}_small_matrices_same_call.jsonl"), (f"soak_perm_{tag}.txt", f"{tag}_soak_perm.txt"),
             (f"chaos_{tag}.txt", f"{tag}_chaos_same_call.txt")):
    p = os.path.join(ROOT, "gpurun_out", a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, b))
print("traffic per launch: %.1f GB (read %.1f, written %.1f)" % ((2 * fetch_kb + write_kb) * 1024 / 1e9, 2 * fetch_kb * 1024 / 1e9, write_kb * 1024 / 1e9))
# which library was measured: its SHA-256 as taken on the GPU box at the start of the evidence call, next to the record of
# the build in the tree (sources / compiler / flags: psoap_amd/build.py)
sha_path = os.path.join(ROOT, "gpurun_out", f"lib_sha256_{tag}.txt")
if os.path.exists(sha_path):
    sha = open(sha_path).read().split()[0]
    rec_path = os.path.join(ROOT, "psoap_amd", "csrc", "libpsoap_gp.so.srchash")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
    with open(os.path.join(dst, f"{tag}_summary.md"), "a") as f:
        f.write(f"\nThe library measured: `psoap_amd/csrc/libpsoap_gp.so`, sha256 `{sha}`, as `python -m psoap_amd.build` produces it from the\n"
                f"committed kernel sources (source hash `{rec.get('sources', '?')}`, `{rec.get('compiler', '?')}`, ladder rung\n"
                f"{rec.get('fallback_rung', '?')}); GPU tests, soak, bench line, these traces and the latency / scheme tables of `profiles/{tag}_*` come from one\n"
                f"call on one box (`tools/evidence_round.sh {tag}`).\n")

