# usage: bash tools/prof_cmd.sh <tag> [bench args...]; kernel statistics of one bench.py run under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --concurrent-steps 0 "$@" > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err || exit 1
python $R/tools/kstats.py $R/gpurun_out/prof_$tag/p_kernel_stats.csv | head -24
