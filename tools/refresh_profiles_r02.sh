#!/bin/bash
# Round-2 profiles of the v2 bench (run on the GPU box from the repo root; writes gpurun_out/prof2/).
# rocprofv3 is run from /tmp with the program itself after `--`; the --pmc passes carry only --kernel-trace.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_line.log 2>$O/bench_line.err
tail -1 $O/bench_line.log > $O/r02_bench_line.json
echo "[prof] bench line done"
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/under_rocprof.log 2>$O/under_rocprof.err
tail -1 $O/under_rocprof.log > $O/r02_bench_under_rocprof.json
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $O/r02_bench_kernel_stats.csv
echo "[prof] kernel stats done"
rm -rf /tmp/pf && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
echo "[prof] FETCH_SIZE pass done"
rm -rf /tmp/pw && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
echo "[prof] WRITE_SIZE pass done"
ALG=$(python3 -c "import json;print(json.load(open('$O/r02_bench_line.json'))['roofline']['algorithmic_bytes_per_launch'])")
python3 $R/tools/mega_traffic.py /tmp/pf /tmp/pw $O/r02_mega_traffic.json $ALG
