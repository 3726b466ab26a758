#!/usr/bin/env python3
"""How the engine's rate depends on which hardware queues its streams land on: creates K HIP streams before the engine
(as another library of the process would) and runs bench.py. Diagnostic.
    python tools/queue_shift_probe.py K [bench.py arguments]"""
import ctypes
import os
import runpy
import sys

import torch

k = int(sys.argv[1])
torch.cuda.set_device(0)
torch.cuda.synchronize()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
streams = []
for _ in range(k):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    streams.append(s)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
