"""Print the measured fp64 MFMA and HBM ceilings of the current GPU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd.chunk import microbench
print(json.dumps(microbench()))

