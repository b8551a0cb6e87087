export TMPDIR=/tmp PYTHONPATH=$PWD
for cfg in "32 256" "0 256" "32 768" "0 768"; do set -- $cfg
  AUDIOCAPTION_DEC_WIDE_MIN=$1 timeout 240 rocprofv3 --kernel-trace --stats -d gpurun_out/p_w -- python tools/decode_wide_bench.py $2 > /dev/null 2>&1
  for f in $(find gpurun_out/p_w -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/wide_stats_$1_$2.txt; done
  rm -rf gpurun_out/p_w
done
head -14 gpurun_out/wide_stats_*.txt
