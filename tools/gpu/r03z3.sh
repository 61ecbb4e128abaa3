# after a last source change: the stored-rows / 16-row parts of the GPU suite, then profiles/traffic.json for the final library
# (two separate PMC passes) and the default bench line with it
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03z3; mkdir -p $O; rm -rf $O/*
CHMC_ROW_SPLIT=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sir or row_split or compact_row or mfma" > $O/pytest_alt.log 2>&1; tail -2 $O/pytest_alt.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
CMD="python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2"
cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1; cd $R
cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1; cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv"); W=$(find $O/pmc_write -name "*counter_collection.csv")
python tools/pmc_summary.py $F $W $O/traffic.json fhn_noisy "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2" > $O/traffic_summary.txt 2>&1; tail -3 $O/traffic_summary.txt
python - $F $W <<'PY'
import sys, pandas as pd
for path, tag in zip(sys.argv[1:], ('fetch', 'write')):
    df = pd.read_csv(path)
    df['kernel'] = df['Kernel_Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 80)
    g = df.groupby(['kernel', 'Counter_Name'])['Counter_Value'].agg(['mean', 'count']).reset_index()
    g.to_csv(f'gpurun_out/r03z3/pmc_{tag}_by_kernel.csv', index=False)
PY
find $O -name "*.csv" -size +4M -delete
cp $O/traffic.json $R/profiles/traffic.json
python bench.py > $O/bench_fhn_noisy_with_traffic.json 2> $O/bench.err || tail -5 $O/bench.err
python bench.py --config sir > $O/bench_sir.json 2> $O/bench_sir.err
python bench.py --config sir --chains-per-gpu 1024 --no-cpu-baseline > $O/bench_sir_1024.json 2> $O/bench_sir_1024.err
python bench.py --engine async --config sir --no-cpu-baseline > $O/bench_sir_engine_async.json 2> $O/bench_sir_async.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_sir -- python3 $R/bench.py --no-cpu-baseline --config sir > $O/prof_stats_sir.log 2>&1; cd $R
cp $(find $O/prof_stats_sir -name "*kernel_stats.csv") $O/kernel_stats_sir.csv
find $O/prof_stats_sir -name "*.csv" -size +4M -delete
python examples/sir_boarding_school_chmc.py 256 300 100 16 > $O/example_sir_boarding_school_256.log 2>&1; tail -2 $O/example_sir_boarding_school_256.log
python tools/adam_timing.py 1024 > $O/adam_init_1024.log 2>&1; tail -1 $O/adam_init_1024.log; python tools/adam_timing.py 256 > $O/adam_init_256.log 2>&1; tail -1 $O/adam_init_256.log
python - <<'PY'
import json
for f in ('bench_fhn_noisy_with_traffic', 'bench_sir', 'bench_sir_1024', 'bench_sir_engine_async'):
    d = json.loads(open(f'gpurun_out/r03z3/{f}.json').read().strip().splitlines()[-1]); r = d['roofline']
    print(f, round(d['value']), round(d['ms_per_step'], 3), r['bound'], round(r['frac'], 3), 'traffic', r.get('traffic'), 'whole', d['config'].get('whole_step_hbm_frac'))
PY
