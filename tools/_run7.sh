export TMPDIR=/tmp PYTHONPATH=$PWD AUDIOCAPTION_TRAFFIC_OPTIONAL=1
for hy in 1 0; do
AUDIOCAPTION_DEC_HYBRID=$hy timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/p_hy -- python bench.py --mode effb2 --steps 10 > /dev/null 2>&1
for f in $(find gpurun_out/p_hy -name "*results.db"); do python profiles/rocpd_summary.py $f > gpurun_out/hy${hy}_effb2_stats.txt; done
rm -rf gpurun_out/p_hy
done
grep -E "dec_|gemm_tiled|gemm_nt|beam_|cache_gather|add_layernorm|attn_step" gpurun_out/hy1_effb2_stats.txt | cut -c1-150
echo ----
grep -E "dec_|gemm_tiled|gemm_nt|beam_|cache_gather|add_layernorm|attn_step" gpurun_out/hy0_effb2_stats.txt | cut -c1-150
