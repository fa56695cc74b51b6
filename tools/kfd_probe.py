import os, sys, glob
sys.path.insert(0, ".")
from psoap_amd import synthetic as syn, covariance
ch = syn.make_chunk(1, 4, 100, seed=1)
covariance.lnlike_f(None, ch.lwls[0], ch.fl, ch.sigma, 0.2, 5.0)
me = os.getpid()
base = "/sys/class/kfd/kfd/proc"
for p in sorted(os.listdir(base)):
    d = os.path.join(base, p)
    try:
        qs = os.listdir(os.path.join(d, "queues"))
    except Exception as e:
        qs = ["ERR " + str(e)[:40]]
    gp = []
    for q in qs[:40]:
        try:
            gp.append(open(os.path.join(d, "queues", q, "gpuid")).read().strip())
        except Exception as e:
            gp.append("?")
    vr = {f: open(os.path.join(d, f)).read().strip() for f in os.listdir(d) if f.startswith("vram_")}
    print(p, "ME" if int(p) == me else "", "queues", len(qs), "gpuids", sorted(set(gp)), "vram", vr)
for n in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*")):
    try:
        gid = open(n + "/gpu_id").read().strip()
        props = dict(l.split() for l in open(n + "/properties").read().splitlines() if len(l.split()) == 2)
        print(n.split("/")[-1], "gpu_id", gid, "location_id", props.get("location_id"), "domain", props.get("domain"), "simd_count", props.get("simd_count"))
    except Exception as e:
        print(n, "ERR", str(e)[:60])
import ctypes
L = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
L.hipDeviceGetPCIBusId(buf, 64, 0)
print("pci bus id of device 0:", buf.value)
