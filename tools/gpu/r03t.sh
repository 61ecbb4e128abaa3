# junction tolerance of the time-parallel scan: statuses / iteration counts against the sequential scan over many chain-steps
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03t; mkdir -p $O; rm -f $O/*
for m in base JTOL3e-13 JTOL1e-12; do
  lib=$R/build/libchmc_$m.so; [ $m = base ] && lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip.so
  for dt in 0.25 0.4; do
    CHMC_HIP_LIBRARY=$lib timeout -k 10 300 python tools/par_scan_compare.py 256 200 $dt 24 > $O/cmp_${m}_$dt.log 2>&1
    echo "$m dt $dt: $(tail -1 $O/cmp_${m}_$dt.log | cut -c1-200); status lines $(grep -c 'status differs' $O/cmp_${m}_$dt.log)"
  done
done
