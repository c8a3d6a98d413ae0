mkdir -p gpurun_out/$1; timeout 1200 python profiles/fuzz_rx_gpu.py ${2:-120} gpurun_out/$1/fuzz_rx_gpu.json 2> gpurun_out/$1/fuzz_rx.err; tail -3 gpurun_out/$1/fuzz_rx.err
