#!/bin/bash
# Round-3 auxiliary profiles (run on the GPU box from the repo root; writes gpurun_out/prof3aux/):
# kernel tables of the v3 flow-matching decoder at BASELINE configs[3]'s chunk shape and of the AR prefill at configs[1]'s shape.
# rocprofv3 is run from /tmp with the program itself after `--`.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof3aux
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/cfm_bench.py --steps 32 > $O/cfm.log 2>$O/cfm.err
tail -1 $O/cfm.log > $O/r03_cfm_bench_32steps.json
echo "[aux] cfm bench done"
rm -rf /tmp/kc && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python3 $R/tools/cfm_bench.py --steps 32 > $O/cfm_under_rocprof.log 2>$O/cfm_under_rocprof.err
tail -1 $O/cfm_under_rocprof.log > $O/r03_cfm_under_rocprof.json
cp $(find /tmp/kc -name '*kernel_stats.csv' | head -1) $O/r03_cfm_kernel_stats.csv
echo "[aux] cfm kernel stats done"
rm -rf /tmp/kp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $R/tools/prefill_probe.py > $O/r03_prefill_probe.txt 2>$O/prefill.err
cp $(find /tmp/kp -name '*kernel_stats.csv' | head -1) $O/r03_prefill_kernel_stats.csv
echo "[aux] prefill kernel stats done"
python3 $R/bench.py > $O/bench_line.log 2>$O/bench_line.err
tail -1 $O/bench_line.log > $O/r03_bench_line.json
echo "[aux] bench line done"
