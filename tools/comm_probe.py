import sys, time, os
sys.path.insert(0, ".")
t0=time.time()
from pysteps_amd import parallel, _lib
from pysteps_amd.device import DeviceArray, synchronize
import numpy as np
_lib.lib()
print("lib ready %.1fs"%(time.time()-t0), flush=True)
t0=time.time()
comm = parallel.Communicator(0, 1, lambda p: p)
print("comm init %.1fs"%(time.time()-t0), flush=True)
d = DeviceArray.from_host(np.ones((1<<20,), np.float32))
t0=time.time(); comm.broadcast(d); synchronize(); print("bcast %.3fs"%(time.time()-t0), flush=True)
t0=time.time(); comm.close(); print("close %.1fs"%(time.time()-t0), flush=True)
