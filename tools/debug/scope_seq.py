"""debug: one oscilloscope random operation sequence (tests/test_gpu_state_machine.py) with the trace comparison printed per step"""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import openmeters_amd
from openmeters_amd.capi import Api
import test_gpu_state_machine as t
omx = openmeters_amd.api()
oracle = Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
seed = int(sys.argv[1])
orig = np.abs
class Spy:
    pass
import builtins
src = open(os.path.join(ROOT, "tests", "test_gpu_state_machine.py")).read()
# run the test body with the last assert replaced by a print
body = src.replace("        assert d[~edge].max(initial=0.0) <= max(0.05, 1.1 * float(smooth.max(initial=0.0))), (seed, step)",
                   "        if d.max() > 0.01: print('step', step, 'max diff', d.max(), 'non-edge max', d[~edge].max(initial=0.0), 'at', np.unravel_index(d.argmax(), d.shape), 'spc', g.samples_per_channel, 'rate', ra, rb, 'frames', frames, 'kind', kind, 'cfg', cfg)\n        if d[~edge].max(initial=0.0) > 0.05:\n            np.save('/tmp/scope_g.npy', g.samples); np.save('/tmp/scope_w.npy', w.samples)")
assert body != src, 'the assert this script replaces has changed'
ns = {}
exec(compile(body, "sm", "exec"), ns)
ns["test_oscilloscope_random_operation_sequences"](omx, oracle, seed)
g, w = np.load('/tmp/scope_g.npy'), np.load('/tmp/scope_w.npy')
print("shapes", g.shape)
g, w = g.reshape(2, -1), w.reshape(2, -1)
d = np.abs(g - w)
for ch in range(2):
    idx = np.argsort(d[ch])[-6:]
    print("channel", ch, "largest diffs at", sorted(idx.tolist()), d[ch][sorted(idx.tolist())])
    i = int(d[ch].argmax())
    print("  g", g[ch, i - 3:i + 4]); print("  w", w[ch, i - 3:i + 4])
    print("  other channel g", g[1 - ch, i - 3:i + 4]); print("  other channel w", w[1 - ch, i - 3:i + 4])
for ch in range(0):
    a, b = g[ch], w[ch]
    best = None
    for sh in range(-4, 5):
        if sh >= 0: d = np.abs(a[sh:] - b[:len(b) - sh]).max()
        else: d = np.abs(a[:sh] - b[-sh:]).max()
        print("channel", ch, "shift", sh, "max diff", d)
