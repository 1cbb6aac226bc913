mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_packed.py tests/test_gpu_timegroups.py tests/test_gpu_parity.py tests/test_gpu_boxplane.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for r in 1 2 3; do
 for L in tools/probes/liblec_old.so ""; do
  for A in "--moving --timesteps 512 --moving-layout cube" "--moving --timesteps 2048 --moving-layout cube"; do
   LEC_LIB=$L python3 bench.py --cpu-baseline none --steps 10 --warmup 3 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-30s %-50s value %9.1f pass ms %7.3f launch ms %7.3f frac %.4f' % ('$L' or 'new', '$A', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
  done
 done
done
