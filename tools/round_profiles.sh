# Development tool: the profile set committed under profiles/ each round (run on the GPU box from the repo root).
set -x
export TMPDIR=/tmp
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
rocprofv3 --kernel-trace --stats -d gpurun_out/p_head -- python bench.py --no-cpu-baseline > gpurun_out/r02_prof_head.json 2>/dev/null
for f in $(find gpurun_out/p_head -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/r02_kernel_stats.txt; done
rm -rf gpurun_out/p_head
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tiers --no-train --no-effb2 --no-steady-state"
mkdir -p gpurun_out/r02_pmc_f16x2
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/p_$c -- $CMD > /dev/null 2>&1
  for f in $(find gpurun_out/p_$c -name "*results.db"); do python profiles/pmc_summary.py $f conv3x3 logmel gru dec_ > gpurun_out/r02_pmc_f16x2/$c.txt; done
  rm -rf gpurun_out/p_$c
done
rocprofv3 --kernel-trace --stats -d gpurun_out/p_train -- python bench.py --mode train --steps 10 > /dev/null 2>&1
for f in $(find gpurun_out/p_train -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/r02_train_kernel_stats.txt; done
rm -rf gpurun_out/p_train
python bench.py --mode train > gpurun_out/r02_train_bench.json 2>/dev/null
python bench.py --mode effb2 > gpurun_out/r02_effb2_bench.json 2>/dev/null
python bench.py --mode effb2 --seconds 30 --beam 4 --effb2-batch 64 > gpurun_out/r02_effb2_30s_beam4.json 2>/dev/null
AUDIOCAPTION_CONV_ALGO=winograd rocprofv3 --kernel-trace --stats -d gpurun_out/p_wino -- python bench.py --no-cpu-baseline --no-tiers --no-train --no-effb2 --steps 10 > gpurun_out/r02_bench_winograd.json 2>/dev/null
for f in $(find gpurun_out/p_wino -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/r02_kernel_stats_winograd.txt; done
rm -rf gpurun_out/p_wino
head -c 600 gpurun_out/r02_bench_final.json; echo; head -14 gpurun_out/r02_kernel_stats.txt | cut -c1-160; head -6 gpurun_out/r02_pmc_f16x2/FETCH_SIZE.txt | cut -c1-200
