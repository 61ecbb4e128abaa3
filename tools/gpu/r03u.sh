# alternative kernel selections under the whole SIR / 16-row part of the GPU suite
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
for cfg in "CHMC_PAR_WAVES=4" "CHMC_PAR_WAVES=1 CHMC_ROW_SPLIT=1" "CHMC_PAR_SCAN=0"; do
  echo "== $cfg" >> $O/alt.log
  env $cfg timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sir or Sir or adam or dynamic or async" >> $O/alt.log 2>&1
  tail -2 $O/alt.log
done
