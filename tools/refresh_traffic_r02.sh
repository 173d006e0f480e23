#!/bin/bash
# PMC traffic of the persistent decode engine alone (the two --pmc passes of tools/refresh_profiles_r02.sh + the reduction)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
echo "[prof] FETCH_SIZE pass done"
rm -rf /tmp/pw && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
echo "[prof] WRITE_SIZE pass done"
ALG=$(python3 -c "import json;print(json.load(open('$R/profiles/r02_bench_line.json'))['roofline']['algorithmic_bytes_per_launch'])")
python3 $R/tools/mega_traffic.py /tmp/pf /tmp/pw $O/r02_mega_traffic.json $ALG
cat $O/r02_mega_traffic.json
