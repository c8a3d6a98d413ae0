mkdir -p gpurun_out/$1; timeout 900 python profiles/fuzz_lmac_frames_gpu.py ${2:-120} gpurun_out/$1/fuzz_lmac_frames_gpu.json 2> gpurun_out/$1/fuzz.err; tail -3 gpurun_out/$1/fuzz.err
