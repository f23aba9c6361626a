for rep in 1 2; do
for lib in "" oldc; do
L=""; [ -n "$lib" ] && L=build/variants/$lib/libpano360_hip.so
PANO_LIB=$L timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('${lib:-new}', 'step', round(d['ms_per_step'],3), 'compose', round(k['multiband_compose_kernel'],3))"
done; done
