# two ranks on the ONE GPU of the box (both on device 0, gloo for the collectives): the multi-rank code path of bench.py with
# the real library under torch.distributed.run, as the driver launches it; then bench.py's own launcher (--gpus 2 without a
# torchrun environment).  RCCL itself needs two GPUs and is not exercised here.
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04ac}; mkdir -p $O; rm -rf $O/*
CHMC_BENCH_DEVICE=0 CHMC_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 > $O/torchrun_2ranks.json 2> $O/torchrun_2ranks.err; echo "rc $?"; tail -c 600 $O/torchrun_2ranks.json; tail -3 $O/torchrun_2ranks.err
CHMC_BENCH_DEVICE=0 CHMC_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 16 --warmup 4 --config sir > $O/selflaunch_2ranks_sir.json 2> $O/selflaunch_2ranks_sir.err; echo "rc $?"; tail -c 300 $O/selflaunch_2ranks_sir.json; tail -3 $O/selflaunch_2ranks_sir.err
O=$O python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.environ['O'] + '/*.json')):
    try:
        d = json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); c = d['config']
        print(os.path.basename(f), d['n_gpus'], round(d['value']), round(d['ms_per_step'], 3), c.get('ranks_joined'), c.get('per_rank_ms'), c.get('collective_backend'), c.get('collective_world_size'), c.get('gathered_sample_shape'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
