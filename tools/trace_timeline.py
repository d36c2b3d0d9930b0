"""One steady-state call of a capture-group trace as a timeline: python tools/trace_timeline.py <t_kernel_trace.csv> [anchor kernel substring]
(rocprofv3 --kernel-trace output).  Calls are cut at the anchor kernel (default: the first kernel of the caller's stream per call)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "scope_push2_kernel"
rows = [r for r in rows if "omx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
mid = idx[len(idx) * 3 // 4]
nxt = idx[len(idx) * 3 // 4 + 1]
# a window: from the earliest kernel of this call to the next anchor
t0 = int(rows[mid]["Start_Timestamp"])
lo = mid
while lo > 0 and int(rows[lo - 1]["End_Timestamp"]) > t0 - 150000 and anchor not in rows[lo - 1]["Kernel_Name"]:
    lo -= 1
base = int(rows[lo]["Start_Timestamp"])
print(f"period between anchors: {(int(rows[nxt]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[lo:nxt + 3]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("omx::", "")[:48]
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    print(f"q{r['Queue_Id']:>2} {s / 1e3:8.1f} -> {e / 1e3:8.1f}  ({(e - s) / 1e3:6.1f} us)  {name}")
