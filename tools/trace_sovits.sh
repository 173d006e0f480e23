#!/bin/bash
# kernel trace of a short bench run grouped by (kernel, grid): which shapes each conv / GEMM kernel is serving
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "dec_gemm" in n or "decode_attn" in n or "sample_step" in n or "t2s_mega" in n: continue
    k=n.replace("_ZN3gsv","").replace("void gsv::","")[:60]
    g=(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X"))
    d[(k,g)].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
out=sorted(d.items(), key=lambda kv:-sum(kv[1]))
for (k,g),v in out[:45]:
    print(f"{k:60s} grid={g} n={len(v)} avg={sum(v)/len(v)/1e3:.1f}us tot={sum(v)/1e6:.2f}ms")
PY
