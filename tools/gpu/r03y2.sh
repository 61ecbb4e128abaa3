# evidence for DESIGN.md section 4 "Round 3, second part": kernel trace summary of the SIR bench's timed part per setting, sweep statistics
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03y2; mkdir -p $O; rm -f $O/*
bash tools/gpu/r03x.sh > $O/sir_trace_summary_default.txt 2>&1
for wv in 1 2 4; do
  echo "== CHMC_PAR_WAVES=$wv" >> $O/par_scan_sweeps.txt
  CHMC_PAR_WAVES=$wv timeout -k 10 200 python tools/par_scan_stats.py 256 200 2>/dev/null | grep traj >> $O/par_scan_sweeps.txt
done
CHMC_ROW_SPLIT=1 CHMC_PAR_WAVES=1 timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_split1_w1.json 2> $O/e.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03y2/bench_sir_split1_w1.json').read().strip().splitlines()[-1]); c = d['config']
print('CHMC_ROW_SPLIT=1 CHMC_PAR_WAVES=1 (stored-rows state evaluation, one wavefront per chain in the scan):', round(d['value']), 'steps/s', round(d['ms_per_step'], 3), 'ms per step, rounds', c['newton_rounds_per_step'])
PY
tail -22 $O/sir_trace_summary_default.txt
