"""waveform bank, 1024 streams x 64 blocks: ms per call as tools/bench_meters.py times it, over 5 and over 30 calls between the two synchronisations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import io
import bench_meters as bm
for reps in (5, 30, 5, 30):
    buf = io.StringIO()
    bm.waveform(reps=reps, out=buf, sizes=(1024,))
    print(f"reps={reps}:", " | ".join(l.split("call:")[1].split("->")[0].strip() for l in buf.getvalue().splitlines() if l.startswith("waveform")))
