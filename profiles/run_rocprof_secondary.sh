# rocprofv3 --kernel-trace --stats summaries of the kernels beside k_fused: the channeliser (config 5), the
# training-sequence search, the burst synchroniser / demultiplexer and the lower-MAC decoder.  Run on the GPU box:
#   gpurun --timeout 900 -- 'sh profiles/run_rocprof_secondary.sh r04_x'
# Results: gpurun_out/prof_out/<tag>_secondary_*.csv (+ the scripts' own JSON lines); copy what is to be kept into profiles/.
set -x
TAG=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out/prof2
rm -rf $O && mkdir -p $O $GRAFT_REPO_ROOT/gpurun_out/prof_out
cd /tmp && export TMPDIR=/tmp
for job in "chan:profiles/measure_chan.py" "scan:profiles/measure_scan.py" "chain:profiles/measure_pipeline.py"; do
    name=${job%%:*}; script=${job#*:}
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- python $GRAFT_REPO_ROOT/$script > $O/$name.log 2>&1
    f=$(find $O/$name -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_secondary_${name}_kernel_stats.csv
    grep '^{' $O/$name.log > $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_secondary_${name}.jsonl
done
cd $GRAFT_REPO_ROOT && head -12 gpurun_out/prof_out/${TAG}_secondary_*_kernel_stats.csv
