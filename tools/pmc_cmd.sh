# usage: bash tools/pmc_cmd.sh <tag>; three counter passes over a short bench.py run, each a run of its own
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
B="$R/bench.py --steps 1 --warmup 1 --cpu-images 0 --concurrent-steps 0"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_fetch -o p -- python $B > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_write -o p -- python $B > $R/gpurun_out/pmc_${tag}_write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -o p -- python $B > $R/gpurun_out/pmc_${tag}_sq.log 2>&1 || exit 1
cd $R
python tools/pmc_summary.py traffic gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/${tag}_pmc_traffic.json 4
python tools/pmc_summary.py sq gpurun_out/pmc_${tag}_sq gpurun_out/${tag}_sq_counters.json
rm -rf gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_sq
