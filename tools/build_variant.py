"""Build libpsoap_gp.so with extra hipcc flags into ab_libs/<name>.so (A/B runs: PSOAP_GP_LIB=$PWD/ab_libs/<name>.so), with the
same assembly gate as the product build.      python tools/build_variant.py sc1 -DPSOAP_SC1_LOADS"""
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import build as B

name, flags = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "ab_libs")
os.makedirs(out, exist_ok=True)
tmp = tempfile.mkdtemp(prefix="psoap_variant_")
try:
    so, asm, rec = B.compile_checked(flags, tmp, verbose=True)
    shutil.copy(so, os.path.join(out, name + ".so"))
    shutil.copy(asm, os.path.join(out, name + ".s"))
    print(os.path.join(out, name + ".so"), rec["flags"], "rung", rec["fallback_rung"])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
