bash tools/profile_round.sh r01 > gpurun_out/prof_r01.log 2>&1
RBS_STATE=dense bash tools/profile_round.sh r01_dense > gpurun_out/prof_r01_dense.log 2>&1
bash tools/sq_profile.sh r01 > gpurun_out/sq_r01.log 2>&1
