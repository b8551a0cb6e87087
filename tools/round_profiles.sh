# Development tool: the profile set committed under profiles/ each round (run on the GPU box from the repo root).
set -x
export TMPDIR=/tmp PYTHONPATH=$PWD AUDIOCAPTION_TRAFFIC_OPTIONAL=1
R=${ROUND:-r06}
T=${TIER:-wino43}
timeout 900 python bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err
cp gpurun_out/bench_details.json gpurun_out/${R}_bench_details.json
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/p_head -- python bench.py --no-cpu-baseline > gpurun_out/${R}_prof_head.json 2>/dev/null
for f in $(find gpurun_out/p_head -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/${R}_kernel_stats.txt; done
rm -rf gpurun_out/p_head
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tiers --no-train --no-effb2 --no-steady-state"
mkdir -p gpurun_out/${R}_pmc_${T}
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/p_$c -- $CMD > /dev/null 2>&1
  for f in $(find gpurun_out/p_$c -name "*results.db"); do python profiles/pmc_summary.py $f conv3x3 block1 logmel gru dec_ > gpurun_out/${R}_pmc_${T}/$c.txt; done
  rm -rf gpurun_out/p_$c
done
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/p_train -- python bench.py --mode train --steps 10 > /dev/null 2>&1
for f in $(find gpurun_out/p_train -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/${R}_train_kernel_stats.txt; done
rm -rf gpurun_out/p_train
timeout 900 python bench.py --mode train > gpurun_out/${R}_train_bench.json 2>/dev/null
timeout 900 python bench.py --mode effb2 > gpurun_out/${R}_effb2_bench.json 2>/dev/null
timeout 900 python bench.py --mode effb2 --seconds 30 --beam 4 --effb2-batch 64 > gpurun_out/${R}_effb2_30s_beam4.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/p_effb2 -- python bench.py --mode effb2 --steps 10 > /dev/null 2>&1
for f in $(find gpurun_out/p_effb2 -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/${R}_effb2_kernel_stats.txt; done
rm -rf gpurun_out/p_effb2
# HBM bytes of the EfficientNet-B2 encoder: 4 encoder forwards in the run (2 warm-up calls, 1 encoder alone, 1 call)
mkdir -p gpurun_out/${R}_pmc_effb2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/p_e$c -- python tools/effb2_bench.py --method greedy --steps 1 > /dev/null 2>&1
  for f in $(find gpurun_out/p_e$c -name "*results.db"); do python profiles/pmc_summary.py $f > gpurun_out/${R}_pmc_effb2/$c.txt; done
  rm -rf gpurun_out/p_e$c
done
python tools/effb2_traffic.py gpurun_out/${R}_pmc_effb2/FETCH_SIZE.txt gpurun_out/${R}_pmc_effb2/WRITE_SIZE.txt 4 128 > gpurun_out/${R}_traffic_effb2.json
AUDIOCAPTION_CONV_ALGO=winograd timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/p_wino -- python bench.py --no-cpu-baseline --no-tiers --no-train --no-effb2 --steps 10 > gpurun_out/${R}_bench_winograd.json 2>/dev/null
for f in $(find gpurun_out/p_wino -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/${R}_kernel_stats_winograd.txt; done
rm -rf gpurun_out/p_wino
head -c 600 gpurun_out/${R}_bench_final.json; echo; head -14 gpurun_out/${R}_kernel_stats.txt | cut -c1-160; head -6 gpurun_out/${R}_pmc_${T}/FETCH_SIZE.txt | cut -c1-200; cat gpurun_out/${R}_traffic_effb2.json
