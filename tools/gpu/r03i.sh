export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03i
mkdir -p $O
CHMC_TIMING_CLASSES=1 timeout -k 10 300 python tools/dynamic_timing.py > $O/dynamic_timing_classes.log 2>&1; cat $O/dynamic_timing_classes.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/dynamic_timing.py > $O/prof.log 2>&1)
cp $(find $O/prof -name "*kernel_stats.csv") $O/dynamic_kernel_stats.csv
find $O/prof -name "*.csv" -size +4M -delete
head -30 $O/dynamic_kernel_stats.csv | cut -c1-160
