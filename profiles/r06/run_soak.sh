mkdir -p gpurun_out/$1; timeout 1500 python profiles/soak_rx_gpu.py ${2:-2000} gpurun_out/$1/soak_rx_gpu.json 2> gpurun_out/$1/soak.err; echo "rc=$?"; tail -2 gpurun_out/$1/soak.err
