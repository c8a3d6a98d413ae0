set -x
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-check"
O=$GRAFT_REPO_ROOT/gpurun_out/prof
rocprofv3 -L > $O/counters_list.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r01 -- $B > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o r01 -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o r01 -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o r01 -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o r01 -- $B > $O/pmc_sq2.log 2>&1
ls -R $O | head -40
# neighbouring stages: kernel trace of the device-resident chain and of the decoder / scan micro-benchmarks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o pipe -- python $GRAFT_REPO_ROOT/profiles/measure_pipeline.py > $O/pipe.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lmac -o lmac -- python $GRAFT_REPO_ROOT/profiles/measure_lmac.py > $O/lmac.log 2>&1
