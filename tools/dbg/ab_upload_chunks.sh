# A/B of rbs_config-less knob RBS_UPLOAD_CHUNKS on the host-pointer legs (run on the GPU box)
for c in ${CHUNKS:-1 2 1 2}; do
RBS_UPLOAD_CHUNKS=$c python bench.py --no-dense-leg --no-f32-leg --no-configs-leg --no-pmc --no-cpu-baseline ${EXTRA:-} --steps 100 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('chunks=$c', *['%s=%.0f'%(k,d[k]) for k in ('host_api_value','host_api_staged_frame_value','host_api_loglikes_only_value','host_api_native_value','tracker_fps_200','tracker_fps_2000','tracker_fps_native_200','tracker_fps_native_2000','tracker_fps_native_pipelined_2000') if k in d])
"
done
