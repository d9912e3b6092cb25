#!/usr/bin/env python
"""Throughput of the other BASELINE.json configurations (bench.py measures configs[2]):
   python tools/bench_configs.py [--steps 5]   -> one line per config, per-launch table with --ops."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

L7 = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8,
          reconstruct_layers=0, pixel_shuffler_filters=1)
CONFIGS = [
    ("C2 dcscn_L8_F96to48 x2, 256 patches", dict(layers=8, filters=96), 256),
    ("C3 dcscn_L12_F196to48 x2, 1024 patches", dict(), 1024),
    ("C4-net dcscn_L12_F196to48 x4, 512 patches", dict(scale=4), 512),
    ("C5 dcscn_L7_F32to8 x4 DS, 1024 patches", dict(L7, scale=4, depthwise_separable=True), 1024),
    ("L7 dcscn_L7_F32to8 x2 (c-DCSCN), 1024 patches", dict(L7), 1024),
    ("L7x3 dcscn_L7_F32to8 x3 (c-DCSCN), 1024 patches", dict(L7, scale=3), 1024),
    ("L7x4 dcscn_L7_F32to8 x4 (c-DCSCN), 1024 patches", dict(L7, scale=4), 1024),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--ops", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--graph", action="store_true", help="option graph_replay: the pass replayed from a hipGraph")
    args = ap.parse_args()
    import torch
    import dcscn_oracle as O
    from dcscn_amd import engine
    for name, flags, n in CONFIGS:
        if args.only and args.only not in name:
            continue
        cfg = O.make_config(**flags)
        eng = engine.Engine(cfg)
        eng.load_weights(O.synthetic_weights(cfg, seed=0))
        s = cfg["scale"]
        x = torch.rand((n, 48, 48, 1), device="cuda") * 255
        x2 = torch.rand((n, 48 * s, 48 * s, 1), device="cuda") * 255
        y = torch.empty_like(x2)
        st = torch.cuda.current_stream().cuda_stream
        if args.graph:
            eng.set_option("graph_replay", 1)
        for _ in range(3):
            eng.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, 48, 48, st)
        if not args.graph:
            eng.set_option("profile", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, 48, 48, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ms = eng.profile()
        ops = eng.ops()
        px = n * 48 * 48
        macs = sum(o["macs_per_lr_pixel"] for o in ops)
        by = sum(o["bytes_per_lr_pixel"] for o in ops)
        print("%-48s %8.3f ms/step  %8.2f LR Mpix/s  %7.2f TFLOP/s  %7.1f GB/s (unfused activation bytes)  kernels %.3f ms"
              % (name, dt * 1e3, px / dt / 1e6, 2 * macs * px / dt / 1e12, by * px / dt / 1e9, sum(ms)), flush=True)
        if args.ops:
            for o, m in zip(ops, ms):
                # a folded tail executes the composite 5x5 conv, not the layers it replaces: the direct-form MACs of those layers over its
                # time would read as a rate above the chip's peak -- the executed MACs are the ones that make a rate
                folded = "(folded)" in o["name"]
                macs_o = o["executed_macs_per_lr_pixel"] if folded else o["macs_per_lr_pixel"]
                print("    %-26s %-11s k%d %4d->%-4d res%d  %8.3f ms  %7.2f TFLOP/s%s  %7.1f GB/s" % (
                    o["name"], o["kernel"], o["kernel_size"], o["in_channels"], o["out_channels"], o["resolution"], m,
                    2 * macs_o * px / (m * 1e-3) / 1e12 if m else 0, " (executed, composite)" if folded else "",
                    o["bytes_per_lr_pixel"] * px / (m * 1e-3) / 1e9 if m else 0))
        eng.close()


if __name__ == "__main__":
    main()
