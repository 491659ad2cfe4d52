#!/bin/bash
# usage: scripts_build_variant.sh NAME -DFOO=1 ...   -> build_variants/NAME.so
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wno-pragma-once-outside-header "$@" -shared \
  -o build_variants/$name.so dbot_ros_amd/csrc/rbsensor_capi.hip -Wl,-rpath,/opt/rocm/lib
