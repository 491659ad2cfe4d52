timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
tools/ab_env.sh "--no-dense-leg" "A=1"
tools/ab_env.sh "--no-dense-leg --update 0" "A=1"
tools/ab_env.sh "--no-dense-leg --mesh m1,m2,m3 --particles 6666 --steps 30" "A=1"
