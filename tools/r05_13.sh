#!/bin/bash
bash tools/ab_lib.sh "--config c1" base:RBS_SPLIT=1 w8:RBS_SPLIT=1 w8u1:RBS_SPLIT=1 w4u4:RBS_SPLIT=1 w4u6:RBS_SPLIT=1 base:RBS_SPLIT=1 base
bash tools/ab_lib.sh "--config c2" base:RBS_SPLIT=1 w8u1:RBS_SPLIT=1
bash tools/ab_lib.sh "--config c3_slice" base:RBS_SPLIT=1 w8u1:RBS_SPLIT=1
