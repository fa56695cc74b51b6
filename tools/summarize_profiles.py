"""Turn the rocprofv3 CSVs collected by tools/profile_round.sh into a small markdown summary."""
import csv, glob, sys, collections, os, re
out, tag = sys.argv[1], sys.argv[2]

def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None      # (the newest: a directory may hold the files of an earlier call)

print(f"# rocprofv3 summary {tag}\n")
print("Command profiled: `python3 bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-extras --no-strong` on one MI355X (ROCm 7.2): 32 walkers x SB2")
print("N=6000 per step.  From round 6 on the headline is again ONE launch of `k_chol_dag<2, false, false, false, 2>` per step (the")
print("template arguments: C, AUG, LAT, STREAM, WPE): every dispatch of that kernel in this run evaluates 32 matrices -- 5 warm-up +")
print("5 timed + 1 + 5 proposals-resident + 1 event-profiled.  (Rounds 4-5: `--mode stream`, one resident dispatch of")
print("`k_chol_dag<2, false, false, true, 2>` per region; still measured beside the headline by the default `bench.py` --")
print("`stream_beside`.)  The k_stream_* / k_mfma_f64_peak / k_tile_engine_bench kernels are the micro-benchmarks behind")
print("`measured_peak` (libpsoap_bench.so).  The second trace (`trace_full`) is the default `bench.py` with its side legs: the staged")
print("step that times k_fill_sym, predict at the retrieve shape (k_chol_dag<3, true, true, false>), the configs[3] strong leg (8")
print("chunks x 32 walkers in one launch: the one long dispatch of the headline kernel), the lnprob(p) and sampler legs.\n")
f = one("trace/**/*kernel_stats.csv")
if f:
    print("## --kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0]
        print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    print()
f = one("trace_full/**/*kernel_stats.csv")
if f:
    print("## --kernel-trace --stats, default bench.py with the side legs\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0]
        if float(r["Percentage"]) >= 0.01:
            print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    print()

def is_stream(name):
    head = name.split("(")[0].replace(" ", "")
    # template arguments <C, AUG, LAT, STREAM, WPE>: the fourth one (rounds 1-3 had three, the start of round 4 four)
    m = re.search(r"k_chol_dag<([^>]*)>", head)
    return bool(m) and len(m.group(1).split(",")) >= 4 and m.group(1).split(",")[3] == "true"


def is_batch(name):
    head = name.split("(")[0].replace(" ", "")
    m = re.search(r"k_chol_dag<([^>]*)>", head)
    if not m:
        return False
    a_ = m.group(1).split(",")
    return len(a_) >= 4 and a_[0] == "2" and a_[1] == "false" and a_[2] == "false" and a_[3] == "false"


F6000 = 6000.0 ** 3 / 3.0 + 2.0 * 6000.0 ** 2
f = one("trace/**/*kernel_trace.csv")
if f:
    brow = sorted((r for r in csv.DictReader(open(f)) if is_batch(r["Kernel_Name"])), key=lambda r: int(r["Start_Timestamp"]))
    if brow:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in brow]
        print("## every dispatch of the launch-per-step kernel `k_chol_dag<2, false, false, false, 2>` (32 evaluations each), from the kernel trace\n")
        print(f"{len(d)} dispatches, duration min / mean / max = {min(d):.3f} / {sum(d) / len(d):.3f} / {max(d):.3f} ms; "
              f"32 x F(6000) / mean = {32 * F6000 / (sum(d) / len(d) * 1e-3) / 1e12:.2f} TFLOP/s = {32 * F6000 / (sum(d) / len(d) * 1e-3) / 1e12 / 78.6:.4f} of 78.6\n")
        print("| # | duration ms | algorithmic TFLOP/s | of 78.6 |")
        print("|---|---|---|---|")
        for i, ms in enumerate(d):
            tf = 32 * F6000 / (ms * 1e-3) / 1e12
            print(f"| {i + 1} | {ms:.3f} | {tf:.2f} | {tf / 78.6:.4f} |")
        print()
    rows = [r for r in csv.DictReader(open(f)) if is_stream(r["Kernel_Name"])]
    if rows:
        print("## every dispatch of the resident (stream) kernel, from the kernel trace\n")
        print("| dispatch | duration ms | evaluations (bench.py: 5 steps x 32) | algorithmic TFLOP/s | of 78.6 |")
        print("|---|---|---|---|---|")
        F = 6000.0 ** 3 / 3.0 + 2.0 * 6000.0 ** 2
        for r in rows:
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            tf = 160 * F / (ms * 1e-3) / 1e12
            print(f"| {r['Dispatch_Id']} | {ms:.3f} | 160 | {tf:.2f} | {tf / 78.6:.3f} |")
        print()


f = one("trace_driver/**/*kernel_trace.csv")
if f:
    import json
    rows = [r for r in csv.DictReader(open(f)) if is_stream(r["Kernel_Name"])]
    line = None
    logp = os.path.join(out, "trace_driver.log")
    if os.path.exists(logp):
        for ln in open(logp):
            if ln.startswith("{") and '"metric"' in ln:
                line = json.loads(ln)
    if line is not None and line.get("config", {}).get("mode") == "dag":
        # the headline is one launch per step: the TIMED REGION of `--steps 20 --warmup 5` = dispatches 6 .. 25 of the
        # launch-per-step kernel in start order (the 5 warm-up steps come first)
        brow = sorted((r for r in csv.DictReader(open(f)) if is_batch(r["Kernel_Name"])), key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in brow]
        steps, warm = int(line.get("steps", 20)), int(line.get("warmup", 5))
        timed = d[warm:warm + steps]
        print("## the driver's command, `python3 bench.py --steps 20 --warmup 5`, under `rocprofv3 --kernel-trace --stats`\n")
        print(f"The headline kernel `k_chol_dag<2, false, false, false, 2>`: {len(d)} dispatches in the run; in start order the first {warm} are")
        print(f"the warm-up steps and the next {steps} the TIMED REGION (32 evaluations each), the later ones the proposals-resident leg, the")
        print("event-profiled step and the side legs (the one long one: configs[3], 256 matrices).\n")
        if timed:
            mean = sum(timed) / len(timed)
            tf = 32 * F6000 / (mean * 1e-3) / 1e12
            print(f"Timed region: {len(timed)} dispatches, duration min / mean / max = {min(timed):.3f} / **{mean:.3f}** / {max(timed):.3f} ms -> "
                  f"32 x F(6000) / mean = **{tf:.2f} TFLOP/s = {tf / 78.6:.4f} of 78.6** (`roofline.frac` of the line must agree: its own "
                  f"event-timed launch took {line.get('roofline', {}).get('avg_launch_ms')} ms).\n")
        rf = line.get("roofline", {})
        sb = line.get("stream_beside") or {}
        print(f"The bench line of this very run (under the profiler): value {line.get('value'):.1f} evals/s, ms_per_step "
              f"{line.get('ms_per_step'):.3f}, roofline.achieved {rf.get('achieved')}, roofline.frac {rf.get('frac')}, "
              f"avg_launch_ms {rf.get('avg_launch_ms')}; the resident launch beside it: {sb.get('evals_per_s')} evals/s "
              f"({sb.get('ratio_to_value')} x).\n")
        rows = []
    if rows:
        print("## the driver's command, `python3 bench.py --steps 20 --warmup 5`, under `rocprofv3 --kernel-trace --stats`\n")
        print("Every dispatch of the resident kernel `k_chol_dag<2, false, false, true, 2>` in that run.  The warm-up's 5 steps are one")
        print("dispatch of 160 evaluations, the TIMED REGION's 20 steps are the next dispatch: 640 evaluations (`stream.matrices` of the")
        print("bench line counts them on the device), the later ones belong to the side legs (lnprob(p) and the sampler through a")
        print("stream).  `roofline.frac` of the line = 640 x F(6000) / that dispatch's duration / 78.6 TFLOP/s.\n")
        print("| dispatch | duration ms | what | algorithmic TFLOP/s | of 78.6 |")
        print("|---|---|---|---|---|")
        F = 6000.0 ** 3 / 3.0 + 2.0 * 6000.0 ** 2
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        # which dispatch is the timed region: the second one of the run (behind the warm-up's) -- and, where the run's own bench
        # line is there, the one whose duration is its roofline.avg_launch_ms (the two must agree)
        big = 1 if len(rows) > 1 else (0 if rows else -1)
        if line is not None and rows:
            want = line.get("roofline", {}).get("avg_launch_ms")
            if want:
                big = min(range(len(rows)), key=lambda i: abs(durs[i] - float(want)))
        for i, r in enumerate(rows):
            ms = durs[i]
            if i == big:
                tf = 640 * F / (ms * 1e-3) / 1e12
                print(f"| {r['Dispatch_Id']} | {ms:.3f} | **the timed region: 640 evaluations** | **{tf:.2f}** | **{tf / 78.6:.4f}** |")
            elif i == 0:
                print(f"| {r['Dispatch_Id']} | {ms:.3f} | the warm-up's 5 steps (160 evaluations) | | |")
            else:
                print(f"| {r['Dispatch_Id']} | {ms:.3f} | a side leg (lnprob(p) through a stream: 6 x 32; the streamed sampler: 32 + 31 x 32) | | |")
        if line is not None:
            rf = line.get("roofline", {})
            print(f"\nThe bench line of this very run (under the profiler): value {line.get('value'):.1f} evals/s, ms_per_step "
                  f"{line.get('ms_per_step'):.3f}, roofline.achieved {rf.get('achieved')}, roofline.frac {rf.get('frac')}, "
                  f"avg_launch_ms {rf.get('avg_launch_ms', rf.get('launch_ms'))}.")
        print()
        f2 = one("trace_driver/**/*kernel_stats.csv")
        if f2:
            print("| kernel (driver's command) | calls | total ms | avg us | % |")
            print("|---|---|---|---|---|")
            for r in csv.DictReader(open(f2)):
                if float(r["Percentage"]) >= 0.05:
                    print(f"| {r['Name'].split('(')[0]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
            print()


def counters(sub):
    f = one(f"{sub}/**/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    if not f:
        return agg, calls, dur
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            calls[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg, calls, dur

agg, calls, dur = counters("pmc_sq")
for k, v in agg.items():
    if "dag" in k:
        n = calls[k]
        print(f"## SQ counters, {k} ({n} dispatches, values per dispatch)\n")
        for c, val in sorted(v.items()):
            print(f"- {c}: {val/n:.4g}")
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8 * 1024      # 8 XCDs summed; 256 CUs x 4 SIMDs
        print(f"- derived: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) = {v['SQ_VALU_MFMA_BUSY_CYCLES']/simd_cycles:.3f}")
        print(f"- derived: executed fp64 MFMA flops per dispatch = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = {v['SQ_INSTS_VALU_MFMA_MOPS_F64']/n*512:.4g}")
        print(f"- derived: mean clock = GRBM_GUI_ACTIVE/8 / duration = {v['GRBM_GUI_ACTIVE']/8/ (dur[k]*1e-6) /1e9:.3f} GHz (avg dispatch {dur[k]/n/1e3:.2f} ms under the profiler)")
        print()
for sub, cname, corr, note in (("pmc_fetch", "FETCH_SIZE", 2.0, "x2 gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM)"),
                               ("pmc_write", "WRITE_SIZE", 1.0, "no correction")):
    agg, calls, dur = counters(sub)
    for k, v in agg.items():
        if "dag" in k and cname in v:
            n = calls[k]
            kb = v[cname] / n
            print(f"## {cname}, {k}: {kb:.4g} KB per dispatch raw -> {kb*1024*corr/1e9:.3f} GB per dispatch ({note}); avg dispatch {dur[k]/n/1e3:.2f} ms\n")
