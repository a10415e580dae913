for l in pathtracer_amd/libprt_hip.so gpurun_in/libprt_hip_sb512.so gpurun_in/libprt_hip_fake256.so gpurun_in/libprt_hip_fake512.so; do
PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --no-cpu-baseline --bounces 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', 'bounces 0', {n: k[n]['ms_per_launch'] for n in k}, flush=True)" || exit 1
done
