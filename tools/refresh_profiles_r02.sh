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
echo "[prof] PMC reduction done"
cd $R
python3 tools/mega_prof.py $O/mega_prof_raw.txt > $O/r02_mega_profile.txt 2>&1
echo "[prof] engine time line done"
python3 tools/sample_bench.py > $O/r02_sampling.json 2>$O/sampling.err
python3 tools/gemm_probe.py 5760x512x1536 5760x512x512 5760x512x2048 5760x2048x512 934x1024x3072 934x1024x1024 934x2048x1024 > $O/r02_gemm_probe.txt 2>&1
python3 tools/conv_probe.py 10 128x128x3x1x512000 128x128x7x1x512000 128x128x11x5x512000 256x256x11x1x64000 64x64x11x5x1024000 16x1x7x1x4096000x2 > $O/r02_conv_probe.txt 2>&1
bash tools/trace_sovits.sh > $O/r02_trace_by_grid.txt 2>&1
echo "[prof] probes done"
# numbers the parity tests print (agreement rates, measured errors): one pass with -s, lines starting with '['
python3 -m pytest tests/test_t2s_mega_gpu.py tests/test_frontend_gpu.py tests/test_t2s_gpu.py tests/test_vits_gpu.py tests/test_cfm_gpu.py -q -m gpu -s 2>&1 | grep -a "\[mega\]\|\[frontend\]\|\[t2s\]\|\[vits\]\|\[cfm\]\|passed\|failed" | sed 's/^\.*//' > $O/r02_parity_log.txt || true
echo "[prof] parity log done"
