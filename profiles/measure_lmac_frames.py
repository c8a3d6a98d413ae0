"""Times tetra_lmac_decode_frames_device (k_lmac_frames) on one second of 4096 channels' frames: every slot a burst, per four slots one
SYNC, one NORM_2 and two NORM_1 (bench.py's coded-downlink mix), random payload.  Variants: all kinds in one launch in the chain's order
(long blocks first), the reverse order, every job in a launch of its own, and the one launch with its grid cut to the rows that exist
(what the empty workgroups of the worst-case grid cost).  HIP events on the launch stream."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
lb, bb = pkg.lmac_binding, pkg.bsync_binding
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
C, F = 4096, 72
n = C * F
pattern = torch.tensor([3, 1, 0, 0], dtype=torch.int32, device=dev)              # SYNC, NORM_2, NORM_1, NORM_1
ft = pattern[(torch.arange(n, device=dev) % F) % 4].contiguous()
frames = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 16), dtype=torch.int64, device=dev, generator=g).to(torch.int32).contiguous()
codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (C,), dtype=torch.int64, device=dev, generator=g).to(torch.int32).repeat_interleave(F).contiguous()
lists = torch.zeros((4, n), dtype=torch.int32, device=dev)
counts = torch.zeros(4, dtype=torch.int32, device=dev)
bb.index_device(ft, F, lists, counts)
torch.cuda.synchronize()
cnt = [int(x) for x in counts.cpu()]
zeros = torch.zeros(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream(dev)
KINDS = {"schf": (5, 0, 1, 288), "sb2": (1, 2, 0, 144), "ndb1": (2, 1, 2, 144), "ndb2": (2, 2, 2, 144), "bbk": (3, 0, 3, 32), "sb1": (0, 1, 0, 80)}


def job(name, tight):
    t, blk, li, stride = KINDS[name]
    return dict(type=t, blk_num=blk, row_frame=lists[li], n_rows=counts[li:li + 1], max_rows=cnt[li] if tight else n, out_stride=stride,
                frame_scramb=codes, type2=torch.zeros((n, stride), dtype=torch.uint8, device=dev), crc_ok=torch.zeros(n, dtype=torch.int32, device=dev),
                labels=torch.zeros((n, 6), dtype=torch.int32, device=dev))


def timed(launches, ws=None, reps=20):
    for _ in range(3):
        for jobs in launches:
            lb.decode_frames_device(frames, ft, jobs, F, zeros, zeros, zeros, stream=s, d_workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        for jobs in launches:
            lb.decode_frames_device(frames, ft, jobs, F, zeros, zeros, zeros, stream=s, d_workspace=ws)
    e1.record(s)
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


order = ["schf", "sb2", "ndb1", "ndb2", "bbk"]
J = {k: job(k, False) for k in KINDS}
Jt = {k: job(k, True) for k in KINDS}
ws = torch.empty(lb.decode_frames_workspace_bytes([J[k] for k in order]), dtype=torch.uint8, device=dev)
res = {"rows": {k: cnt[KINDS[k][2]] for k in KINDS},
       "one_launch_long_first_ms": timed([[J[k] for k in order]], ws),
       "one_launch_short_first_ms": timed([[J[k] for k in reversed(order)]], ws),
       "one_launch_tight_grid_ms": timed([[Jt[k] for k in order]], ws),
       "one_launch_pool_scratch_ms": timed([[J[k] for k in order]], None),
       "launch_per_job_ms": {k: timed([[J[k]]], ws) for k in order + ["sb1"]},
       "coded_only_one_launch_ms": timed([[J[k] for k in order[:4]]], ws)}
# SCH/F alone with exactly w working waves (64 rows each), tight grid: how a launch scales from one wave per SIMD (1024) upwards
scal = {}
for w in (256, 1024, 2048, 3072, 4096, 8192):
    j = dict(Jt["schf"], max_rows=min(64 * w, cnt[1]), n_rows=None)
    scal[str(min(w, (cnt[1] + 63) // 64))] = timed([[j]], ws)
res["schf_ms_by_working_waves"] = scal
res["launch_per_job_sum_ms"] = round(sum(res["launch_per_job_ms"][k] for k in order), 4)
print(json.dumps(res))
