#!/bin/bash
# Round-3 profiles of the v2 bench (run on the GPU box from the repo root; writes gpurun_out/prof3/).
# rocprofv3 is run from /tmp with the program itself after `--`; the --pmc passes carry only --kernel-trace.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_line.log 2>$O/bench_line.err
tail -1 $O/bench_line.log > $O/r03_bench_line.json
echo "[prof] bench line done"
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/under_rocprof.log 2>$O/under_rocprof.err
tail -1 $O/under_rocprof.log > $O/r03_bench_under_rocprof.json
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $O/r03_bench_kernel_stats.csv
echo "[prof] kernel stats done"
rm -rf /tmp/pf && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
echo "[prof] FETCH_SIZE pass done"
rm -rf /tmp/pw && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
echo "[prof] WRITE_SIZE pass done"
ALG=$(python3 -c "import json;print(json.load(open('$O/r03_bench_line.json'))['roofline']['algorithmic_bytes_per_launch'])")
python3 $R/tools/mega_traffic.py /tmp/pf /tmp/pw $O/r03_mega_traffic.json $ALG
echo "[prof] PMC reduction done"
cd $R
python3 tools/mega_prof.py $O/mega_prof_raw.txt > $O/r03_mega_profile.txt 2>&1
PROF_OFF=1 PROF_B=64 python3 tools/mega_prof.py 2>/dev/null | grep mode > $O/r03_mega_batch_sweep.txt
PROF_OFF=1 PROF_B=96 python3 tools/mega_prof.py 2>/dev/null | grep mode >> $O/r03_mega_batch_sweep.txt
PROF_OFF=1 PROF_B=128 python3 tools/mega_prof.py 2>/dev/null | grep mode >> $O/r03_mega_batch_sweep.txt
PROF_OFF=1 PROF_B=128 GSV_MEGA_QUADS=seq python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/^/sequential quads: /' >> $O/r03_mega_batch_sweep.txt
echo "[prof] engine time line + batch sweep done"
python3 tools/sample_bench.py > $O/r03_sampling.json 2>$O/sampling.err
echo "[prof] sampling done"
# numbers the parity tests print (agreement rates, measured errors): one pass with -s, lines starting with '['
python3 -m pytest tests/test_t2s_engine_parity_gpu.py tests/test_t2s_mega_gpu.py tests/test_full_size_configs_gpu.py tests/test_pipeline_v3_gpu.py tests/test_vits_gpu.py -q -m gpu -s 2>&1 | grep -a "^\.*\[\|passed\|failed" | sed 's/^\.*//' | cut -c1-400 > $O/r03_parity_log.txt || true
echo "[prof] parity log done"
