for lib in mitsuba_amd/libmtsamd.so variants/lib_sb128.so variants/lib_sb64.so; do for r in 8 16; do
MTSAMD_LIB=$PWD/$lib MTSAMD_SHADE_ROUNDS=$r timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/v.json 2>/dev/null || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/v.json').read()); k=d['kernel_ms_rank0']; print('$lib', $r, round(d['value'],1), {x: round(k[x]/d['steps'],1) for x in ('extend','shade','shadow')})"
done; done
