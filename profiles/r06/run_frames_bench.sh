mkdir -p gpurun_out/$1
timeout 300 python profiles/measure_lmac_frames.py > gpurun_out/$1/measure_lmac_frames.json 2> gpurun_out/$1/measure_lmac_frames.err; cat gpurun_out/$1/measure_lmac_frames.json; tail -2 gpurun_out/$1/measure_lmac_frames.err
