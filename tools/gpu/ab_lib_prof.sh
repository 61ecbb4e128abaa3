# kernel averages (rocprofv3 --kernel-trace --stats) of the default bench for the shipped library and for build/libchmc_<variant>.so,
# then the interleaved bench A/B of ab_lib.sh        usage: ab_lib_prof.sh <tag> <variant> <kernel name pattern>
export TMPDIR=/tmp
R=$PWD
TAG=$1; VAR=$2; PAT=$3
O=$R/gpurun_out/$TAG; mkdir -p $O
for lib in shipped $VAR; do
  L=""; [ $lib = shipped ] || L=$R/build/libchmc_$VAR.so
  (cd /tmp && env ${L:+CHMC_HIP_LIBRARY=$L} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --repeats 0 > $O/prof_$lib.log 2>&1)
  echo "== $lib"; grep -h "$PAT" $(find $O/prof_$lib -name "*kernel_stats.csv") | cut -c1-150; rm -rf $O/prof_$lib
done
bash tools/gpu/ab_lib.sh ${TAG}_ab $VAR
