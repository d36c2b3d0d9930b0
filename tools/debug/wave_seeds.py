"""run tests/test_gpu_parity_meters.py::test_waveform_chunk_parallel_random_sequences on the given seeds and print every three-way
bar's largest ratio (HIP library: OMX_HIP_LIB or the product)"""
import os, sys
root = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import conftest, parity
import test_gpu_parity_meters as m
import openmeters_amd
from openmeters_amd.capi import Api

omx = openmeters_amd.api()
oracle = Api(conftest._build_oracle(), "omxo_")
for seed in [int(x) for x in sys.argv[1:]]:
    parity.LEDGER.clear()
    try:
        m.test_waveform_chunk_parallel_random_sequences(omx, oracle, seed)
        verdict = "green"
    except AssertionError as e:
        verdict = "RED " + str(e)[:300]
    print(seed, verdict)
    for name, (limit, worst, n) in sorted(parity.LEDGER.items()):
        if "random sequences" in name:
            print(f"   {worst:10.3e}  ({n:4d} checks)  {name}")
