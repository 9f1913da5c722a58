"""One rank through the slab path with a REAL RCCL communicator (UDC_FORCE_SLAB=1 UDC_FORCE_COMM=1: every exchange is an
ncclSend / ncclRecv to itself on the communication stream) and four k-chunks, a few substeps of a neutral channel with the floor --
run under rocprofv3 --kernel-trace by profiles/tools/pipe_trace.sh to see which compute kernels RCCL's kernels run beside."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
from udcore.core import DynCore      # noqa: E402
from udcore.grid import Grid         # noqa: E402

torch.cuda.set_device(0)
nx, ny, nz = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 256, 256)
g = Grid.uniform(nx, ny, nz)
core = DynCore(g, sgs=2, nsv=0, lbottom=True, z0=0.05)
buf = (ctypes.c_ubyte * 128)()
assert core.lib.udc_comm_unique_id(buf) == 0
core.comm_init(bytes(buf))
core.set_forcing(np.full(nz, -1e-4), np.zeros(nz))
rng = np.random.default_rng(5)
for k, base in (("u0", 1.0), ("v0", 0.0), ("w0", 0.0)):
    a = np.zeros(g.mshape())
    a[1:-1, 1:-1, 1:-1] = base + 0.04 * (rng.random((nz, ny, nx)) - 0.5)
    if k == "w0":
        a[1] = 0.
    core.upload(k, a); core.upload(k.replace("0", "m"), a)
core.halos(); core.boundary()
import time
core.run(3, 0.25)
core.sync()
nsub = int(os.environ.get("PIPE_TRACE_SUBSTEPS", "6"))
t0 = time.perf_counter()
core.run(nsub, 0.25)
core.sync()
print("ms_per_substep", (time.perf_counter() - t0) / nsub * 1e3, "divmax", core.divergence()[0])
core.close()
