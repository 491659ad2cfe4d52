#!/bin/bash
# round 5: everything the driver runs at round end (GPU suite, smoke, default bench) + the round's profiles
out=gpurun_out/r05_full; mkdir -p $out
( time python -m pytest tests -m gpu -x -q ) > $out/gputests.log 2>&1; echo "gpu suite rc=$?"; tail -6 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $out/smoke.log
( time python bench.py ) > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; tail -c 600 $out/bench_default.json; tail -3 $out/bench_default.err
bash tools/profile_all.sh r05 > $out/profile_all.log 2>&1; echo "profiles rc=$?"; ls gpurun_out/profiles_r05 | head -30
