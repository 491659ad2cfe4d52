for round in 1 2; do
for e in "X=1" "RBS_COPY_TPB=64" "RBS_COPY_TPB=128" "RBS_COPY_TPB=64 RBS_COPY_ROWS=4" "RBS_COPY_TPB=64 RBS_COPY_ROWS=1" "RBS_COPY_TPB=128 RBS_COPY_ROWS=1"; do
  env $e python bench.py --quick --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e] round=$round step_ms=%.4f copy_ms=%.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
