# Everything the round's report cites, for the library build in the tree (run as ONE gpurun call, <= 20 min):
#   counter traffic of the four single-GPU workloads -> profiles/traffic.json (+ per-kernel means), default bench line with
#   cpu baseline and other_configs, kernel statistics (rocprofv3 --kernel-trace --stats) of the headline and of SIR, SIR
#   bench lines (256 / 512 / 1024 chains; 8 / 4 wavefronts per chain and batched launches forced), phase breakdown of the
#   per-chain kernels, Adam finder timings.     usage: evidence.sh <tag>
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r04}
O=$R/gpurun_out/${TAG}_evidence; mkdir -p $O; rm -rf $O/*
P=$O/profiles; mkdir -p $P   # (only gpurun_out/ travels back: copy $P/* into profiles/ afterwards)
bash tools/gpu/pmc_traffic.sh $TAG > $O/pmc.log 2>&1 || { tail -20 $O/pmc.log; exit 1; }
tail -4 $O/pmc.log
cp $R/profiles/traffic.json $R/profiles/${TAG}_pmc_*_by_kernel.csv $P/
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
cp $O/bench_default.json $P/${TAG}_bench_default.json
for cfg in sir fhn_noiseless; do
  WU=4; if [ $cfg = sir ]; then WU=16; fi   # (SIR: the timed region starts at a trajectory boundary: one launch per trajectory)
  timeout -k 10 300 python bench.py --config $cfg --warmup $WU --no-other-configs > $O/bench_$cfg.json 2> $O/e.log || tail -5 $O/e.log
  cp $O/bench_$cfg.json $P/${TAG}_bench_$cfg.json
done
timeout -k 10 300 python bench.py --config sir --warmup 16 --chains-per-gpu 1024 --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_sir_1024.json 2> $O/e.log || tail -5 $O/e.log
CHMC_RETRACT_KERNEL=0 timeout -k 10 300 python bench.py --config sir --warmup 16 --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_sir_lockstep_rounds.json 2> $O/e.log || tail -5 $O/e.log
CHMC_RETRACT_KERNEL=0 timeout -k 10 300 python bench.py --config sir --warmup 16 --chains-per-gpu 1024 --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_sir_1024_lockstep_rounds.json 2> $O/e.log || tail -5 $O/e.log
timeout -k 10 300 python bench.py --config sir --warmup 16 --chains-per-gpu 512 --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_sir_512.json 2> $O/e.log || tail -5 $O/e.log
for b in 256 512 1024; do for k in 1 2; do
  CHMC_RETRACT_KERNEL=$k timeout -k 10 300 python bench.py --config sir --warmup 16 --chains-per-gpu $b --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_sir_${b}_wavefronts_$((k == 1 ? 8 : 4)).json 2> $O/e.log || tail -5 $O/e.log
done; done
timeout -k 10 300 python bench.py --num-steps-per-obs 800 --chains-per-gpu 512 --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_fhn_noisy_s800_512.json 2> $O/e.log || tail -5 $O/e.log
timeout -k 10 300 python bench.py --solver quasi-newton --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_fhn_quasi_newton.json 2> $O/e.log || tail -5 $O/e.log
timeout -k 10 300 python bench.py --splitting gaussian --no-cpu-baseline --no-other-configs > $P/${TAG}_bench_fhn_gaussian.json 2> $O/e.log || tail -5 $O/e.log
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --repeats 0 > $O/prof.log 2>&1)
cp $(find $O/prof -name "*kernel_stats.csv") $P/${TAG}_kernel_stats.csv; rm -rf $O/prof
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sir -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --repeats 0 --config sir > $O/prof_sir.log 2>&1)
cp $(find $O/prof_sir -name "*kernel_stats.csv") $P/${TAG}_kernel_stats_sir.csv; rm -rf $O/prof_sir
if [ -f $R/build/libchmc_prof.so ]; then
  CHMC_HIP_LIBRARY=$R/build/libchmc_prof.so timeout -k 10 300 python tools/retract_prof.py 256 2 > $P/${TAG}_sir_phase_breakdown_256.txt 2>&1
fi
python tools/adam_timing.py 1024 0 > $P/${TAG}_adam_init_1024.log 2>&1
python tools/adam_timing.py 1024 0 variable > $P/${TAG}_adam_init_1024_variable_sigma.log 2>&1
python - $P $TAG <<'PY'
import json, sys, glob, os
P, TAG = sys.argv[1:3]
for f in sorted(glob.glob(f'{P}/{TAG}_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']; r = d['roofline']
        print(os.path.basename(f), round(d['value']), round(d['ms_per_step'], 3), r['kernel'], r['bound'], round(r['frac'], 3), 'traffic', r.get('traffic'),
              'whole', c.get('whole_step_hbm_frac'), 'rep', (c.get('value_repeats') or {}).get('values'))
        for k, v in (c.get('other_configs') or {}).items():
            print('    ', k, v.get('value') and round(v['value']), v.get('ms_per_step') and round(v['ms_per_step'], 3), v.get('error'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
