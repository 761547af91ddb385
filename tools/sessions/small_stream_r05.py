"""How long does a plain streaming read of N MB take (k_stream_read, 4096 workgroups, back to back)?  The floor under the level-1 passes (70 MB, 24 us) and
the other small kernels of the coarse part."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip.device import Context
c = Context(0)
for mb in (8, 16, 32, 64, 70, 128, 256, 512, 1024, 3072):
    gbs = c.membench(mb << 20)
    print("%5d MB: %7.1f us per pass, %7.1f GB/s" % (mb, (mb << 20) / gbs / 1e3, gbs), flush=True)
