"""Durations of the window-fold kernels (window_sum_kernels.hip) over shapes, through the test hook omx_debug_window_sums; run under
rocprofv3 --kernel-trace --stats (tools/prof_fold.sh) to read the kernel times."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openmeters_amd

api = openmeters_amd.api()
f = api.fn("debug_window_sums", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p])
rng = np.random.default_rng(1)
shapes = [(64, 4096, 256, 1024), (64, 4096, 256, 256), (64, 1024, 256, 1024), (64, 4096, 1024, 256), (64, 16384, 1024, 256), (16, 4096, 256, 1024), (256, 4096, 256, 256)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in sys.argv[1].split(","))]
for S, W, hop, n_hops in shapes:
    need = W + hop * (n_hops - 1)
    cap = 1 << int(np.ceil(np.log2(need + 1)))
    ring = rng.uniform(-1, 1, (S, cap)).astype(np.float32)
    out = np.zeros((S, n_hops), np.float32)
    for _ in range(3):
        api.check(f(ring.ctypes.data, S, cap, 0, hop, W, n_hops, out.ctypes.data))
    print(S, W, hop, n_hops, "cap", cap, "checksum", float(out.sum()))
