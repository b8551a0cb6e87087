# Development tool: PMC passes over tools/conv_bench.py for one conv algo / layer set (run on the GPU box from the repo root).
#   bash tools/w1_pmc.sh <algo> <layers> <tag>
export TMPDIR=/tmp PYTHONPATH=$PWD
ALGO=${1:-wino1d}; LAYERS=${2:-b3c2,b6c2}; TAG=${3:-w1}
CMD="python tools/conv_bench.py --algos $ALGO --layers $LAYERS --iters 3"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/p_$i -- $CMD > /dev/null 2>&1
  for f in $(find gpurun_out/p_$i -name "*results.db"); do python profiles/pmc_summary.py $f conv3x3 > $OUT/pass$i.txt; done
  rm -rf gpurun_out/p_$i
done
cat $OUT/pass*.txt | cut -c1-170
