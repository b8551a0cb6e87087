set -x
export TMPDIR=/tmp PYTHONPATH=$PWD
R=r05
python bench.py --mode effb2 > gpurun_out/${R}_effb2_bench.json 2>/dev/null
python bench.py --mode effb2 --seconds 30 --beam 4 --effb2-batch 64 > gpurun_out/${R}_effb2_30s_beam4.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/p_effb2 -- python bench.py --mode effb2 --steps 10 > /dev/null 2>&1
for f in $(find gpurun_out/p_effb2 -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/${R}_effb2_kernel_stats.txt; done
rm -rf gpurun_out/p_effb2
mkdir -p gpurun_out/${R}_pmc_effb2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/p_e$c -- python tools/effb2_bench.py --method greedy --steps 1 > /dev/null 2>&1
  for f in $(find gpurun_out/p_e$c -name "*results.db"); do python profiles/pmc_summary.py $f > gpurun_out/${R}_pmc_effb2/$c.txt; done
  rm -rf gpurun_out/p_e$c
done
python tools/effb2_traffic.py gpurun_out/${R}_pmc_effb2/FETCH_SIZE.txt gpurun_out/${R}_pmc_effb2/WRITE_SIZE.txt 4 128 > gpurun_out/${R}_traffic_effb2.json
cat gpurun_out/${R}_effb2_bench.json | tail -1 | cut -c1-400; cat gpurun_out/${R}_traffic_effb2.json | tail -8
