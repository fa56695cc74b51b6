"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and per-launch TFLOP/s of k_panel_update."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].split('(')[0]
    tot[name][0] += 1; tot[name][1] += d
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} calls {n:5d} total {us/1e3:9.2f} ms avg {us/n:9.1f} us")
pu = [r for r in rows if 'k_panel_update' in r['Kernel_Name']]
if pu:
    gy = max(int(r['Grid_Size_Y']) for r in pu)
    last = [r for r in pu if int(r['Grid_Size_Y']) == gy][-46:]
    print("panel_update launches of the last full-batch step (p, tiles, us, executed TF):")
    for r in last:
        gx = int(r['Grid_Size_X']) // 256
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        P = 47; p = P - gx; k0 = 128 * p
        fl = 2 * 128 * 128 * k0 * gx * gy
        print(f"  p={p:2d} tiles={gx:2d} blocks={gx*gy:5d} {d:8.0f} us {fl/d/1e6:6.1f} TF")
