mkdir -p gpurun_out/r06o
./profiles/microbench/int_rate2.bin > gpurun_out/r06o/int_rate2.jsonl
wc -l gpurun_out/r06o/int_rate2.jsonl
