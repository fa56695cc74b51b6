"""Turn the rocprofv3 CSVs collected by tools/profile_round.sh into a small markdown summary."""
import csv, glob, sys, collections, os
out, tag = sys.argv[1], sys.argv[2]

def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None

print(f"# rocprofv3 summary {tag}\n")
print("Command profiled: `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg4` on one MI355X (ROCm 7.2): 32 walkers x SB2")
print("N=6000 per step.  k_chol_dag dispatches: 1 warm-up + 3 timed + 1 event-profiled + 3 PCIe-inclusive + 4 through the")
print("lnprob(p) boundary + 4 driven by the multi-chain MH sampler; the staged kernels (k_panel_update / k_potrf_diag / k_trsm_strip / k_fill_sym) come from the one")
print("staged step bench.py runs to time the stand-alone fill kernel; the k_stream_* / k_mfma_f64_peak / k_tile_engine_bench")
print("kernels are the micro-benchmarks behind `measured_peak`.\n")
f = one("trace/**/*kernel_stats.csv")
if f:
    print("## --kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0]
        print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
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
