# HBM traffic per kernel class by counters for every single-GPU BASELINE workload, for the library build in the tree:
# two SEPARATE rocprofv3 --pmc passes per workload (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md), merged into
# profiles/traffic.json (keyed by workload, SHA-256 of the profiled library) by tools/pmc_summary.py, plus the per-kernel
# means under profiles/<tag>_pmc_<key>_by_kernel.csv.       usage: bash tools/gpu/pmc_traffic.sh <tag> [keys...]
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r04}; shift
KEYS=${@:-"fhn_noisy sir fhn_noisy_s800_b512 fhn_noiseless"}
O=$R/gpurun_out/${TAG}_pmc; mkdir -p $O
for key in $KEYS; do
  case $key in
    fhn_noisy) ARGS="";;
    fhn_noiseless) ARGS="--config fhn_noiseless";;
    sir) ARGS="--config sir";;
    fhn_noisy_s800_b512) ARGS="--config fhn_noisy --num-steps-per-obs 800 --chains-per-gpu 512";;
    *) echo "unknown workload key $key"; exit 1;;
  esac
  SW="--steps 4 --warmup 2"
  if [ $key = sir ]; then SW="--steps 16 --warmup 16"; fi  # (whole trajectories: one launch of k_traj_chain each)
  CMD="python3 $R/bench.py --no-cpu-baseline --no-other-configs --repeats 0 $SW $ARGS"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/$key.$ctr
    (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$key.$ctr -- $CMD > $O/$key.$ctr.log 2>&1) || { echo "pmc pass $key $ctr failed"; tail -5 $O/$key.$ctr.log; exit 1; }
  done
  F=$(find $O/$key.FETCH_SIZE -name "*counter_collection.csv"); W=$(find $O/$key.WRITE_SIZE -name "*counter_collection.csv")
  python tools/pmc_summary.py $F $W $R/profiles/traffic.json $key "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-configs --repeats 0 $SW $ARGS" > $O/$key.summary.txt 2>&1; tail -3 $O/$key.summary.txt
  python - $F $W $R/profiles/${TAG}_pmc_${key}_by_kernel.csv <<'PY'
import sys, pandas as pd
rows = []
for path in sys.argv[1:3]:
    df = pd.read_csv(path)
    df['kernel'] = df['Kernel_Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 80)
    rows.append(df.groupby(['kernel', 'Counter_Name'])['Counter_Value'].agg(['mean', 'count']).reset_index())
pd.concat(rows).to_csv(sys.argv[3], index=False)
PY
  rm -rf $O/$key.FETCH_SIZE $O/$key.WRITE_SIZE
done
cp $R/profiles/traffic.json $O/traffic.json
