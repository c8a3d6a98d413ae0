# rocprofv3 evidence for the bench line (run on the GPU box through gpurun): one --kernel-trace --stats pass and the counter
# passes, each --pmc run on its own (never combined with other trace domains), then profiles/pmc_summary.py condenses them.
#   gpurun --timeout 1500 -- 'sh profiles/run_rocprof.sh r02_x [channels per GPU, default 4096]'
# Results land in gpurun_out/prof_out/ ; copy the ones to keep into profiles/.
set -x
TAG=${1:-r03}
CH=${2:-4096}
O=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $O && mkdir -p $O $GRAFT_REPO_ROOT/gpurun_out/prof_out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-check --no-host-path --no-large-batch --no-config5 --no-time-major --no-chain --channels $CH"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r02 -- $B > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o r02 -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o r02 -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o r02 -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o r02 -- $B > $O/pmc_sq2.log 2>&1
cd $GRAFT_REPO_ROOT && python profiles/pmc_summary.py $O gpurun_out/prof_out $TAG $CH 36000
tail -3 $O/trace.log
cat gpurun_out/prof_out/${TAG}_rocprof_summary.md
