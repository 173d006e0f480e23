#!/bin/bash
# A/B of the fused K/V scatter + V^T prefill launch (default) against the two launches (GSV_PREFILL_SPLIT_SCATTER=1):
# kernel tables of tools/prefill_probe.py under rocprofv3.  Run on the GPU box from the repo root; writes gpurun_out/ab_kvt/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_kvt
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ka && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka -- python3 $R/tools/prefill_probe.py > $O/fused.txt 2>$O/fused.err
cp $(find /tmp/ka -name '*kernel_stats.csv' | head -1) $O/fused_kernel_stats.csv
export GSV_PREFILL_SPLIT_SCATTER=1
rm -rf /tmp/kb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -- python3 $R/tools/prefill_probe.py > $O/split.txt 2>$O/split.err
cp $(find /tmp/kb -name '*kernel_stats.csv' | head -1) $O/split_kernel_stats.csv
unset GSV_PREFILL_SPLIT_SCATTER
{ echo "fused (default):"; tail -1 $O/fused.txt; grep -a "prefill_kvt\|prefill_vt\|kv_scatter" $O/fused_kernel_stats.csv | cut -d, -f1-4;
  echo "split (GSV_PREFILL_SPLIT_SCATTER=1):"; tail -1 $O/split.txt; grep -a "prefill_kvt\|prefill_vt\|kv_scatter" $O/split_kernel_stats.csv | cut -d, -f1-4; } > $O/r03_ab_prefill_kvt.txt
cat $O/r03_ab_prefill_kvt.txt
