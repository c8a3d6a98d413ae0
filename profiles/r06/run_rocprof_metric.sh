# rocprofv3 evidence of the metric kernel (profiles/run_rocprof.sh), raw output removed afterwards
sh profiles/run_rocprof.sh ${1:-r06_y} > gpurun_out/run_rocprof.log 2>&1; tail -8 gpurun_out/run_rocprof.log
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
