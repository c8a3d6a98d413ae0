#!/usr/bin/env python3
"""Channel-count sweep of the hot path on ONE GPU: launch time of k_fused for C channels x 36000 samples,
input resident in HBM, steady clocks (same ramp as bench.py).  One JSON line per C.
    python profiles/sweep_channels.py [--channels 256 800 4096 8192 16384 32768] [--steps 6]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[256, 800, 4096, 8192, 16384, 32768])
    ap.add_argument("--samples", type=int, default=36000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--shape", choices=["auto", "narrow", "wide", "small"], default="auto",
                    help="workgroup shape: chosen by the library from the channel count, or forced (16 / 32 / 4 channels per workgroup)")
    ap.add_argument("--layout", choices=["channel_major", "time_major"], default="channel_major")
    a = ap.parse_args()
    import torch
    import tetra_amd
    import bench
    pkg = tetra_amd.pkg
    dev = torch.device("cuda", 0)
    clk_khz, cus = pkg.binding.device_info(0)
    N = a.samples
    stride = pkg.binding.bits_stride(N)
    base = None
    for C in a.channels:
        iq, _ = bench.make_input(torch, pkg.synth, dev, C, N, seed=20260000)
        bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev)
        nb = torch.zeros(C, dtype=torch.int32, device=dev)
        flags = {"auto": 0, "narrow": pkg.binding.FLAG_NARROW_WORKGROUPS, "wide": pkg.binding.FLAG_WIDE_WORKGROUPS,
                 "small": pkg.binding.FLAG_SMALL_WORKGROUPS}[a.shape]
        tm = a.layout == "time_major"
        if tm:
            iq = iq.transpose(0, 1).contiguous()
        dem = pkg.Demodulator(C, N, flags=flags, layout=pkg.binding.LAYOUT_TIME_MAJOR if tm else pkg.binding.LAYOUT_CHANNEL_MAJOR)
        st = torch.cuda.current_stream(dev)
        for _ in range(bench.RAMP_STEPS + 2):
            dem.process_device(iq, N, bits, stride, nb, None, st)
        torch.cuda.synchronize()
        for _ in range(a.steps):
            dem.process_device(iq, N, bits, stride, nb, None, st)
        torch.cuda.synchronize()
        k1 = dem.kernel_ms_history(a.steps)
        ms = float(k1.mean())
        wgs = (C + 15) // 16
        if C == 4096:
            base = ms
        print(json.dumps({"channels": C, "samples": N, "shape": a.shape, "layout": a.layout, "kernel_ms": round(ms, 4), "workgroups_of_16": wgs, "cus": cus,
                          "workgroups_of_16_per_cu": round(wgs / cus, 2), "msamples_s": round(C * N / ms / 1e3, 1),
                          "ms_per_4096_channels": round(ms * 4096 / C, 4),
                          "vs_4096": round(ms / base, 3) if base else None}), flush=True)
        dem.close()
        del iq, bits, nb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
