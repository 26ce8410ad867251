timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
run() { env "${@:2}" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-counters --no-d2h-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', d['value'],d['ms_per_step'])"; }
run default
run s1 RTMI_STREAMS=1
run s3 RTMI_STREAMS=3
