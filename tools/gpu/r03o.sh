export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
for wv in 1 2 4; do
  echo "== W=$wv" >> $O/stats.log
  CHMC_PAR_WAVES=$wv timeout -k 10 200 python tools/par_scan_stats.py 256 200 >> $O/stats.log 2>&1
done
cat $O/stats.log
