# round 3, part b: new GPU parity tests, FETCH_SIZE calibration, latency-floor ubench output, bench lines with the hardware roofline
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03b
mkdir -p $O
timeout -k 10 120 tools/ubench/fwd_latency.bin > $O/fwd_latency_ubench.txt 2>&1; echo "fwd_latency rc $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -- $R/tools/ubench/fetch_calib.bin > $O/calib.log 2>&1); echo "calib rc $?"
python tools/fetch_calib.py $(find $O/calib -name "*counter_collection.csv" | head -1) $O/fetch_calibration.json
find $O/calib -name "*.csv" -size +4M -delete
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_fhn_noisy.json 2> $O/bench_fhn_noisy.err || tail -5 $O/bench_fhn_noisy.err
python bench.py --config sir > $O/bench_sir.json 2> $O/bench_sir.err || tail -5 $O/bench_sir.err
echo done
