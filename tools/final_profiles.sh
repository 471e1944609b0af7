# usage (on the GPU box): bash tools/final_profiles.sh <tag>
# One unprofiled bench line and one rocprofv3 kernel-statistics run per workload, plus the encoder's per-layer tables; everything
# lands under gpurun_out/ with <tag> in the name, ready to be copied into profiles/.
R=$GRAFT_REPO_ROOT
tag=$1
cd $R
for wl in transformer lstm swin; do
  timeout -k 10 280 python bench.py --workload $wl > gpurun_out/${tag}_${wl}_bench_line.json 2> gpurun_out/${tag}_${wl}_bench.err || exit 1
  python tools/showbench.py gpurun_out/${tag}_${wl}_bench_line.json || exit 1
  bash tools/prof_cmd.sh ${tag}_${wl} --workload $wl || exit 1
done
timeout -k 10 200 python tools/igemm_sweep.py > gpurun_out/${tag}_encoder_layers.txt 2>&1 || exit 1
timeout -k 10 200 python tools/fused_sweep.py > gpurun_out/${tag}_fused_blocks.txt 2>&1 || exit 1
tail -5 gpurun_out/${tag}_fused_blocks.txt
