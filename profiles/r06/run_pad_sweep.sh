mkdir -p gpurun_out/$1
for pad in 0 2048 4096 5632 8192 12288 16384; do
  TETRA_EXP_LMAC_LDS_PAD=$pad timeout 300 python profiles/measure_lmac_frames.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($pad, d['one_launch_long_first_ms'], d['one_launch_short_first_ms'], d['coded_only_one_launch_ms'], d['launch_per_job_ms']['schf'])"
done | tee gpurun_out/$1/pad_sweep.txt
