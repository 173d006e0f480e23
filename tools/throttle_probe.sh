cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python -c "import torch;print(torch.get_num_threads())"
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
for t in default 1; do
  if [ $t = 1 ]; then export OMP_NUM_THREADS=1 MKL_NUM_THREADS=1; fi
  for i in 1 2 3; do timeout -k 10 150 python bench.py --no-cpu-baseline > gpurun_out/thr_${t}_$i.log 2>&1 && python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], max(d['step_ms']), sorted(d['step_ms'])[-3:])" gpurun_out/thr_${t}_$i.log; grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; done
done
