#!/bin/bash
# The in-suite soak (tests/test_gpu_soak.py) on several seed bases in a row (run on the GPU box): bash tools/soak_suite.sh <first base> <count>
# Per base: the pytest verdict, and every parity exemption the base NEEDED (tests/parity.py: rule, needed / consulted) — a base that is
# green on the plain bars prints only its verdict.  The last line sums the exemptions over the run.
FIRST=${1:-1000}; COUNT=${2:-5}
REP=$(mktemp)
TOTAL=$(mktemp)
for b in $(seq $FIRST $((FIRST + COUNT - 1))); do
  OMX_SOAK_SEED=$b OMX_PARITY_REPORT=$REP python -m pytest tests/test_gpu_soak.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|AssertionError" | head -8 | sed "s/^/base $b: /"
  grep "^exemption:" $REP | awk '{ n = NF; if ($(n-2) + 0 > 0) print }' | sed "s/^/base $b: /"
  grep -E "^exemption:|exempted columns" $REP >> $TOTAL
done
python3 - "$TOTAL" <<'PY'
import re, sys, collections
used, seen, arb = collections.Counter(), collections.Counter(), 0.0
for line in open(sys.argv[1]):
    m = re.match(r"exemption: (.*?)\s+(\d+) /\s+(\d+)\s*$", line)
    if m:
        used[m.group(1)] += int(m.group(2)); seen[m.group(1)] += int(m.group(3))
    elif "exempted columns" in line:
        arb = max(arb, float(line.split()[-3]))
for k in seen:
    print(f"total: {k}: needed {used[k]} of {seen[k]} checks")
print(f"total: worst arbitration |HIP - exact f64| / max(fixed bar, 2 |oracle - exact f64|) = {arb:.3f}")
PY
