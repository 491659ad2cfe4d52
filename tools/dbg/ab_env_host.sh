# A/B of environment settings on the host-pointer and tracker legs (run on the GPU box):
#   bash tools/dbg/ab_env_host.sh "RBS_AUX_DEFER=0" "RBS_AUX_DEFER=1" ...
for e in "$@"; do
env $e python bench.py --no-dense-leg --no-f32-leg --no-configs-leg --no-pmc --no-cpu-baseline ${EXTRA:-} --steps 100 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$e]', *['%s=%.0f'%(k.replace('tracker_fps','tfps').replace('host_api','host'),d[k]) for k in ('host_api_value','host_api_staged_frame_value','host_api_native_value','tracker_fps_200','tracker_fps_2000','tracker_fps_pipelined_2000','tracker_fps_native_200','tracker_fps_native_2000','tracker_fps_native_pipelined_2000') if k in d])
"
done
