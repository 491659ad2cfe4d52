#!/bin/bash
out=gpurun_out/r05k; mkdir -p $out
bash tools/ab_lib.sh "--config c1" base base:RBS_STP_ENTER=0.0 base:RBS_STP_ENTER=0.0,RBS_STP_EVERY=1000000 base > $out/ab.log 2>&1; cat $out/ab.log
bash tools/ab_lib.sh "--config c2" base base:RBS_STP_ENTER=0.0 >> $out/ab2.log 2>&1; cat $out/ab2.log
