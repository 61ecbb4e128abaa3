# round 3 final-build measurements, part 2: HBM traffic (two separate PMC passes), SQ counters, S = 800 traffic, examples
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03z
mkdir -p $O
CMD="python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2"
cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1; cd $R
cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1; cd $R
cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $CMD > $O/pmc_sq.log 2>&1; cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv"); W=$(find $O/pmc_write -name "*counter_collection.csv"); Q=$(find $O/pmc_sq -name "*counter_collection.csv")
python tools/pmc_summary.py $F $W $O/traffic.json fhn_noisy "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2" > $O/traffic_summary.txt 2>&1; tail -15 $O/traffic_summary.txt
CMD8="python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --num-steps-per-obs 800 --chains-per-gpu 512"
cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_s800 -- $CMD8 > $O/pmc_fetch_s800.log 2>&1; cd $R
cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_s800 -- $CMD8 > $O/pmc_write_s800.log 2>&1; cd $R
F8=$(find $O/pmc_fetch_s800 -name "*counter_collection.csv"); W8=$(find $O/pmc_write_s800 -name "*counter_collection.csv")
python - $F $W $Q $F8 $W8 <<'PY'
import sys, pandas as pd
for path, tag in zip(sys.argv[1:], ('fetch', 'write', 'sq', 'fetch_s800_512', 'write_s800_512')):
    df = pd.read_csv(path)
    df['kernel'] = df['Kernel_Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 80)
    g = df.groupby(['kernel', 'Counter_Name'])['Counter_Value'].agg(['mean', 'count']).reset_index()
    g.to_csv(f'gpurun_out/r03z/pmc_{tag}_by_kernel.csv', index=False)
    if tag == 'sq':
        p = g.pivot(index='kernel', columns='Counter_Name', values='mean')
        p['valu_per_wave'] = p['SQ_INSTS_VALU'] / p['SQ_WAVES']
        p['active_frac'] = p['SQ_ACTIVE_INST_ANY'] / p['SQ_WAVE_CYCLES']
        print(p.sort_values('SQ_WAVE_CYCLES', ascending=False).head(12).to_string())
PY
find $O -name "*.csv" -size +4M -delete
cp $O/traffic.json $R/profiles/traffic.json
python bench.py > $O/bench_fhn_noisy_with_traffic.json 2> $O/bench_fhn_noisy_with_traffic.err || tail -5 $O/bench_fhn_noisy_with_traffic.err
python examples/fhn_notebook_posterior.py 64 700 200 24 > $O/notebook_posterior_64x700.log 2>&1; tail -12 $O/notebook_posterior_64x700.log
python examples/fhn_noisy_chmc.py 256 400 120 40 $O/fhn_run > $O/example_fhn_sampler_256x400.log 2>&1; tail -4 $O/example_fhn_sampler_256x400.log
python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic > $O/example_fhn_dynamic_256x400.log 2>&1; grep "leapfrog steps/s" $O/example_fhn_dynamic_256x400.log; python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic-shared > $O/example_fhn_dynamic_shared_step_256x400.log 2>&1; grep "leapfrog steps/s" $O/example_fhn_dynamic_shared_step_256x400.log; python tools/adam_timing.py 1024 > $O/adam_init_1024.log 2>&1; tail -1 $O/adam_init_1024.log; python tools/adam_timing.py 256 > $O/adam_init_256.log 2>&1; tail -1 $O/adam_init_256.log; python tools/dynamic_timing.py > $O/dynamic_timing.log 2>&1; head -3 $O/dynamic_timing.log
python examples/sir_boarding_school_chmc.py 256 300 100 16 > $O/example_sir_boarding_school_256.log 2>&1; tail -4 $O/example_sir_boarding_school_256.log
rm -rf $O/fhn_run
