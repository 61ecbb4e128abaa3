export TMPDIR=/tmp
O=$PWD/gpurun_out/r04g; mkdir -p $O; rm -rf $O/*
timeout -k 10 600 python -m pytest tests/test_hip_surface.py -m gpu -x -q > $O/pytest_surface.log 2>&1 || { tail -40 $O/pytest_surface.log; exit 1; }
tail -2 $O/pytest_surface.log
python tools/adam_timing.py 1024 0 variable > $O/adam_1024_varsigma.log 2>&1; tail -2 $O/adam_1024_varsigma.log
python tools/adam_timing.py 1024 0 > $O/adam_1024.log 2>&1; tail -1 $O/adam_1024.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --collective cabi --steps 16 > $O/bench_cabi.json 2> $O/e.log || tail -5 $O/e.log
python -c "
import json; d = json.loads(open('gpurun_out/r04g/bench_cabi.json').read().strip().splitlines()[-1]); c = d['config']
print(round(d['value']), c['collective_route'], c['cabi_comm_world_size'], c['gathered_sample_shape'])"
