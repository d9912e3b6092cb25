#!/usr/bin/env python
"""Benchmark of the MI355X-native DCSCN forward pass (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the default
``dcscn_L12_F196to48_NIN_A64_PS`` x2 graph on synthetic 48x48 Y-channel patches -- 1024 PER GPU (weak
scaling, the default) or, with ``--strong``, 1024 in total split into contiguous shards of 1024 / N per
rank (512 / 256 / 128 per GPU at 2 / 4 / 8).  Independent patches shard across ranks with
no data-path collective.  A step is one forward pass of the rank's whole batch with x / x2 already
resident in HBM and y left in HBM.  Weights are seeded synthetic (the trained L12 blobs are not shipped).
The graph executed is the library default (linear tail folded into one 5x5 conv, include/dcscn.h
"fold_linear_tail"); the layer-by-layer graph is timed beside it at N = 1 (``layer_by_layer``).

Arithmetic: the library default ("split16", include/dcscn.h) runs the 3x3 stack and the wide 1x1 convs on the f16 matrix
pipe at f32 accuracy -- every f32 operand as an f16 (hi, lo) pair, three products per MAC, f32 accumulation (measured error
below the f32 kernels', profiles/r03_f16x3_numerics.txt); everything else is f32.  The pure-f32 kernels (split16 = 0) are
timed in the same line (`f32_path`).  With N > 1 the line also carries the strong-scaling leg of BASELINE configs[2]
(`strong_scaling`: 1024 patches in total, 1024 / N per rank).

Rank 0 prints one JSON line: LR Mpixels/s over all GPUs, plus
  roofline      -- the dominant kernel (the 3x3 convs: conv3_h on v_mfma_f32_16x16x32_f16, or conv_wino2 on f32 MFMA with
                   split16 = 0): `achieved` = the FLOPs the kernel really issues per step / its summed launch time, measured
                   with HIP events on the launch stream inside the timed steps; `frac` = achieved / dense MFMA peak of the
                   instruction's dtype = matrix-pipe utilisation (<= 1); `useful_frac` leaves out channel padding.  The
                   direct-form (algorithmic, f32-equivalent) FLOP rate of the same launches is reported against the f32 peak
                   (`vs_f32_peak`).  `traffic` and `sustained_clock_GHz` are REPLAYED from the committed rocprofv3 PMC passes of
                   this command (profiles/r*_pmc_per_dispatch.json): PMC counters cannot be read in-process.
  cpu_baseline  -- the float32 torch-CPU restatement of the same graph (oracle/cpu_path_torch.py;
                   TensorFlow is not installable) timed on the host cores on a bounded sample.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATCH = 48
PATCHES_PER_GPU = 1024
MODEL_FLAGS = dict()          # defaults of helper/args.py = dcscn_L12_F196to48_NIN_A64_PS_R1F32, scale 2
MODEL_NAME = "dcscn_L12_F196to48_NIN_A64_PS"
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (at 2.4 GHz)
SUSTAINED_F16_MFMA_TFLOPS = 2100.0   # measured (tools/mfma_two_waves.hip): 1024 SIMDs x 16384 FLOP / 8.0 ns with the whole chip busy
PEAK_HBM_GBS = 8000.0


def _kernel_family(name):
    """Kernel family of a profiler kernel name ('void dcscn::conv3_h8<6, 5, ...>(...)' / 'conv3_h8<6,5,0,6,true>') or of a
    dcscn_op_info kernel: the bare identifier in front of the template arguments."""
    n = name.split("(")[0].split("<")[0].strip()
    n = n.split("::")[-1].split(" ")[-1]
    return KERNEL_ALIASES.get(n, n)


# launches that dcscn_op_info reports under another kernel's name: the border ring of a whole-tail fold (csrc/conv5_h.hpp: fold_border)
# is the second launch of its conv5_h op
KERNEL_ALIASES = {"fold_border": "conv5_h", "conv_nin_h_w8": "conv_nin_h"}      # (conv_nin_h_w8: the 256-pixel workgroups of the wide K axes)


def _pmc_file(prefix, run_kernels=None):
    """Latest committed PMC summary of the bench command in which kernels named ``prefix``... did real work (in a split16
    profile the conv_wino2 / conv_nin launches are the empty fallbacks behind the f16 kernels).  ``run_kernels``: the kernel
    families dcscn_op_info reports for THIS run; a file whose working kernels are a different set was taken on other kernels
    (a kernel changed and tools/rocprof_bench.sh was not re-run) and is refused -- returns (None, "stale: ...")."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_per_dispatch.json")), reverse=True):
        try:
            with open(path) as f:
                kernels = json.load(f)["kernels"]
        except (OSError, KeyError, ValueError):
            continue
        if any(n.startswith(prefix) and k.get("avg_duration_ns_profiled", 0) > 1e5 for n, k in kernels.items()):
            if run_kernels is not None:
                # the kernels that did real work under the profiler (the gated float32 plan's launches exit at once: < 20 us)
                worked = {_kernel_family(n) for n, k in kernels.items() if k.get("avg_duration_ns_profiled", 0) > 2e4}
                worked -= PMC_IGNORED_KERNELS
                want = set(run_kernels) - PMC_IGNORED_KERNELS
                if worked != want:
                    return None, "stale: %s was taken on kernels %s, this run launches %s" % (
                        os.path.basename(path), sorted(worked), sorted(want))
            return path, None
    return None, "no committed PMC file for kernels named %s*" % prefix


# helper kernels that are not launches of the plan (dcscn_op_info does not list them)
PMC_IGNORED_KERNELS = {"pass_begin_kernel", "fill_kernel", "p16_pack_kernel", "p16_unpack_kernel"}


DOM_PREFIX = ["conv3_h"]      # kernel-name prefix of the dominant launches in the PMC file (set from the run: conv3_h / conv_wino)


def csrc_digest():
    """sha256 over the kernel sources (csrc/*, sorted): written into the PMC summary by tools/summarize_rocprof.py and compared
    here, so that a replayed counter can be told from one taken on this very build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "dcscn-super-resolution_amd", "csrc", "*"))):
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def pmc_replay(dom_ms, nin_ms, algorithmic_bytes, run_kernels=None):
    """REPLAYED (not measured in this run): figures from the committed rocprofv3 PMC passes of this same command
    (tools/rocprof_bench.sh -> profiles/r*_pmc_per_dispatch.json), combined with this run's kernel times.
    Returns (traffic, north_star).  traffic: HBM bytes per step of the dominant kernel, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads on gfx950 (the counter tallies 128-B requests at
    64 B), WRITE_SIZE as reported."""
    path, why = _pmc_file(DOM_PREFIX[0], run_kernels)
    if not path:
        return None, {"replayed": False, "traffic": None, "reason": why}
    try:
        with open(path) as f:
            doc = json.load(f)
        kernels = doc["kernels"]
        fetch = write = 0.0
        for name, k in kernels.items():
            if name.startswith(DOM_PREFIX[0]) and "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                per_forward = k["dispatches"] / max(k.get("forwards", 4), 1)
                fetch += k["FETCH_SIZE"] * 1024.0 * per_forward
                write += k["WRITE_SIZE"] * 1024.0 * per_forward
        traffic = None
        if fetch:
            corrected = 2.0 * fetch + write
            traffic = {"bytes_per_step": corrected, "replayed": True, "source": os.path.basename(path),
                       "fetch_bytes_reported": fetch, "write_bytes_reported": write,
                       "correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE as reported",
                       "algorithmic_bytes_per_step": algorithmic_bytes,
                       "traffic_ratio": round(corrected / algorithmic_bytes, 3) if algorithmic_bytes else None}
        ns = {"replayed": True, "source": os.path.basename(path)}
        # counters taken on exactly this build of csrc/ ?  (None: the file predates the digest)
        same_build = (doc["csrc_sha256"] == csrc_digest()) if "csrc_sha256" in doc else None
        ns["counters_taken_on_this_build_of_csrc"] = same_build
        if traffic:
            traffic["counters_taken_on_this_build_of_csrc"] = same_build
        if traffic and dom_ms > 0:
            gbs = traffic["bytes_per_step"] / (dom_ms * 1e-3) / 1e9
            alg_gbs = algorithmic_bytes / (dom_ms * 1e-3) / 1e9
            ns["hbm_3x3_stack"] = {"achieved_GBps": round(gbs, 1), "algorithmic_GBps": round(alg_gbs, 1), "peak_GBps": PEAK_HBM_GBS,
                                   "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_frac": round(alg_gbs / PEAK_HBM_GBS, 4),
                                   "note": "counter bytes (corrected) and SURVEY 8(d) algorithmic bytes over this run's kernel time.  The "
                                           "north_star's >= 0.5 of the HBM roofline is unreachable for this stack in f32-equivalent "
                                           "arithmetic: 24.1 GB at 8 TB/s is 3.0 ms, the three f16 products per MAC alone are 7.3 ms at the "
                                           "nominal 2.5 PFLOP/s (ceiling ~0.41), 10.3 ms at the 1.78 PFLOP/s the pipe sustains with "
                                           "random operands (profiles/r04_mfma_operands.txt: ceiling ~0.29); the stack is MFMA bound"}
        for name, k in kernels.items():
            if (name.startswith("conv_nin") or name.startswith("conv_igemm<1,")) and "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k:
                active = k["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
                if DOM_PREFIX[0] == "conv3_h" and not name.startswith("conv_nin_h"):
                    continue                                  # the f32 fallback launch behind conv_nin_h (empty)
                ns["nin_1x1"] = {"kernel": name, "mfma_util": round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * active), 4),
                                 "sustained_clock_GHz": round(active / k["avg_duration_ns_profiled"], 3) if k.get("avg_duration_ns_profiled") else None,
                                 "lds_bank_conflict_frac": round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"], 4) if k.get("SQ_LDS_IDX_ACTIVE") else None,
                                 "ms_per_step": round(nin_ms, 4),
                                 "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles), fused B1+A1 GEMM"}
            if name.startswith(DOM_PREFIX[0]) and "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k:
                active = k["GRBM_GUI_ACTIVE"] / 8.0
                ns.setdefault("conv_3x3", {})[name] = {
                    "mfma_util": round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * active), 4),
                    "sustained_clock_GHz": round(active / k["avg_duration_ns_profiled"], 3) if k.get("avg_duration_ns_profiled") else None,
                    "lds_bank_conflict_frac": round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"], 4)
                    if k.get("SQ_LDS_IDX_ACTIVE") else None}
        return traffic, ns
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None, None


L7_FLAGS = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8,
                reconstruct_layers=0, pixel_shuffler_filters=1)
# the other single-GPU configurations of BASELINE.json ("also C2, C5"), timed beside the headline at N = 1
OTHER_CONFIGS = {
    "C2": ("dcscn_L8_F96to48 x2 forward, 256 48x48 Y patches (BASELINE.json configs[1])", dict(layers=8, filters=96), 256),
    "C5": ("dcscn_L7_F32to8 x4 depthwise-separable forward, 1024 48x48 Y patches (BASELINE.json configs[4])",
           dict(L7_FLAGS, scale=4, depthwise_separable=True), 1024),
}
PEAK_F32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 32 FMA lanes per clock and SIMD (a wave64 instruction issues over 2 cycles)
C3H_KERNELS = ("conv3_h", "conv3_h8", "conv3_h2")


def other_config_traffic(key, run_kernels, families):
    """REPLAYED HBM bytes per step of an other_configs leg, from the committed PMC passes of tools/rocprof_other_configs.sh
    (profiles/r*_<c2|c5>_pmc_per_dispatch.json; FETCH_SIZE x 2 + WRITE_SIZE as for the headline), summed over the kernel
    `families` (None = every kernel of the pass).  Refused like the headline's when the file was taken on other kernels."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_pmc_per_dispatch.json" % key.lower())), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            kernels = doc["kernels"]
        except (OSError, KeyError, ValueError):
            continue
        worked = {_kernel_family(n) for n, k in kernels.items() if k.get("avg_duration_ns_profiled", 0) > 2e4} - PMC_IGNORED_KERNELS
        if worked != set(run_kernels) - PMC_IGNORED_KERNELS:
            return None, {"traffic": None, "reason": "stale: %s was taken on kernels %s, this run launches %s" % (
                os.path.basename(path), sorted(worked), sorted(run_kernels))}
        total = 0.0
        for n, k in kernels.items():
            if (families is None or _kernel_family(n) in families) and "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                total += (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 * k["dispatches"] / max(k.get("forwards", 6), 1)
        return total, {"bytes_per_step": total, "replayed": True, "source": os.path.basename(path),
                       "correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE as reported",
                       "counters_taken_on_this_build_of_csrc": (doc["csrc_sha256"] == csrc_digest()) if "csrc_sha256" in doc else None}
    return None, {"traffic": None, "reason": "no committed PMC file for %s" % key}


def narrow_net_work(O, cfg):
    """Per LR pixel, from the reference's graph (oracle.build_topology = DCSCN.py:222-325): MACs of the depthwise halves of
    separable convs (VALU work: no contraction over channels), MACs of everything else (matrix work), values entering a
    contraction (each is split into an f16 (hi, lo) pair: 1.5 VALU operations) and values leaving one (scale, PReLU: 3)."""
    res = {"x": 1, "x2": cfg["scale"]}
    dw = mm = vin = vout = 0
    for op in O.build_topology(cfg):
        kind = op["op"]
        if kind == "conv":
            r = res[op["src"]]
            res[op["dst"]] = r
            px = r * r
            k2 = op["k"] * op["k"]
            if op["ds"] and op["k"] > 1:
                dw += px * k2 * op["cin"]
                mm += px * op["cin"] * op["cout"]
            else:
                mm += px * k2 * op["cin"] * op["cout"]
            vin += px * op["cin"]
            vout += px * op["cout"]
        elif kind == "concat":
            res[op["dst"]] = res[op["srcs"][0]]
        elif kind == "depth_to_space":
            res[op["dst"]] = res[op["src"]] * op["block"]
        elif kind == "conv_transpose":
            res[op["dst"]] = res[op["src"]] * op["scale"]
        elif "dst" in op:
            res[op["dst"]] = res.get(op.get("src"), cfg["scale"])
    return {"depthwise_macs": dw, "matrix_macs": mm, "values_in": vin, "values_out": vout}


def time_other_config(engine, O, torch, key, steps, warmup, device_index, stream):
    """One more BASELINE configuration, timed like the headline (inputs resident in HBM, y left in HBM, barrier-free at N = 1:
    synchronize on both sides of exactly `steps` forwards) with the per-launch HIP-event profile of the same steps."""
    name, flags, n = OTHER_CONFIGS[key]
    cfg = O.make_config(**flags)
    eng = engine.Engine(cfg, device=device_index)
    try:
        eng.load_weights(O.synthetic_weights(cfg, seed=0))
        s = cfg["scale"]
        gen = torch.Generator(device="cuda")
        gen.manual_seed(4321)
        x = torch.rand((n, PATCH, PATCH, 1), device="cuda", generator=gen) * 255.0
        x2 = torch.rand((n, PATCH * s, PATCH * s, 1), device="cuda", generator=gen) * 255.0
        y = torch.empty_like(x2)
        torch.cuda.synchronize()

        def fwd():
            eng.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, PATCH, PATCH, stream)
        for _ in range(max(warmup, 3)):
            fwd()
        eng.set_option("profile", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fwd()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        per_op = eng.profile(with_float32_plan=True)
        f32_plan_ms = per_op.pop()
        eng.set_option("profile", 0)
        if not bool(torch.isfinite(y).all().item()):
            raise RuntimeError("non-finite output")
        ops = eng.ops()
        px = n * PATCH * PATCH
        ms = el / steps * 1e3
        out = {"workload": name, "ms_per_step": round(ms, 4), "value": round(px / (ms * 1e-3) / 1e6, 3), "unit": "LR Mpix/s",
               "steps": steps, "kernel_ms_per_step": round(sum(per_op), 4), "float32_plan_gated_ms": round(f32_plan_ms, 4),
               "launches": [{"name": o["name"], "kernel": o["kernel"], "ms": round(m, 4)} for o, m in zip(ops, per_op)]}
        dom = [(o, m) for o, m in zip(ops, per_op) if o["kernel"] in C3H_KERNELS and o["kernel_size"] == 3 and o["out_channels"] > 1]
        if dom:
            # the headline's accounting: f16 FLOPs the 3x3 launches issue (3 products per MAC, channel padding included) per second of
            # their own launch time against the dense f16 MFMA peak; useful = 3 x the algorithmic FLOPs; algorithmic = SURVEY 8(d)
            dms = sum(m for _, m in dom)
            ex = sum(2.0 * o["executed_macs_per_lr_pixel"] for o, _ in dom) * px
            alg = sum(2.0 * o["macs_per_lr_pixel"] for o, _ in dom) * px
            out["roofline"] = {"kernel": "+".join(sorted({o["kernel"] for o, _ in dom})) + " 3x3, %d launches/pass" % len(dom), "bound": "mfma",
                               "achieved": round(ex / (dms * 1e-3) / 1e12, 3), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ex / (dms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                               "useful_frac": round(3.0 * alg / (dms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                               "algorithmic_frac": round(alg / (dms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                               "kernel_ms_per_step": round(dms, 4), "traffic": None}
            tr, detail = other_config_traffic(key, {o["kernel"] for o in ops}, set(C3H_KERNELS))
            out["roofline"]["traffic"] = tr
            out["roofline"]["traffic_detail"] = detail
            if tr:
                by = sum(float(o["bytes_per_lr_pixel"]) for o, _ in dom) * px
                detail["algorithmic_bytes_per_step"] = by
                detail["traffic_ratio"] = round(tr / by, 3) if by else None
        else:
            # streamed narrow net: nothing but x, x2 and y has to touch HBM -- 4 + 4 s^2 + 4 s^2 bytes per LR pixel
            io = 4.0 * (1 + 2 * s * s)
            w = narrow_net_work(O, cfg)
            t_hbm = io * px / (PEAK_HBM_GBS * 1e9) * 1e3
            t_mfma = 3.0 * 2.0 * w["matrix_macs"] * px / (PEAK_F16_MFMA_TFLOPS * 1e12) * 1e3
            valu_ops = w["depthwise_macs"] + 1.5 * w["values_in"] + 3.0 * w["values_out"]
            t_valu = valu_ops * px / (PEAK_F32_VALU_TFLOPS / 2.0 * 1e12) * 1e3
            out["roofline"] = {"kernel": "+".join(o["kernel"] for o in ops), "bound": "hbm",
                               "achieved": round(io * px / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(io * px / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                               "compulsory_bytes_per_lr_pixel": io,
                               "floor_ms": {"hbm": round(t_hbm, 4), "mfma_3_products": round(t_mfma, 4), "valu": round(t_valu, 4),
                                            "valu_plus_mfma": round(t_valu + t_mfma, 4)},
                               "frac_of_floor": round(max(t_hbm, t_valu + t_mfma) / ms, 4),
                               "work_per_lr_pixel": w,
                               "note": "achieved = compulsory I/O (x, x2, y: %d B per LR pixel) over the step; the floor is not HBM but the "
                                       "arithmetic: depthwise MACs + 1.5 per split input value + 3 per output value on the VALU at the "
                                       "spec rate (32 lanes per clock and SIMD, no packed f32) plus three f16 products per matrix MAC at "
                                       "the dense f16 peak, without counting any overlap between the two pipes" % int(io)}
            tr, detail = other_config_traffic(key, {o["kernel"] for o in ops}, None)
            out["roofline"]["traffic"] = tr
            out["roofline"]["traffic_detail"] = detail
            if tr:
                detail["compulsory_bytes_per_step"] = io * px
                detail["traffic_ratio"] = round(tr / (io * px), 3)
                detail["note"] = ("launches %s: Concat2 (32 channels, 128 B per LR pixel) is written by the feature launch and read by the tail's "
                                  "(r06: the whole tail folded into one 5x5 conv, conv5_h + the border ring's fold_border) -- the counters see "
                                  "x + Concat2 and Concat2 (+ the ring's windows) + x2 + y" % "+".join(o["kernel"] for o in ops))
        return out
    finally:
        eng.close()


def parity_leg():
    """The second half of BASELINE's metric, computed in this run: the shipped c-DCSCN x2 checkpoint (tests/golden/weights_L7_x2.npz,
    the reference's trained weights) on the five Set5 images through model.do_for_evaluate (DCSCN.py:672-703) -- PSNR per image against
    the float64 oracle's (tests/golden/goldens.json, written by tests/golden/make_golden.py) -- and the 48x48 crop vector's max-abs
    error against the oracle's float64 output.  Fixtures only: nothing under oracle/ runs here."""
    import tempfile
    from dcscn_amd.model import SuperResolution
    from helper import args as hargs
    golden = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(golden, "goldens.json")) as f:
        g = json.load(f)

    class _F(dict):
        __getattr__ = dict.__getitem__
    d = {n: hargs.FLAGS._flags[n].default for n in hargs.FLAGS}
    d.update(g["models"]["L7_x2"]["flags"])
    d.pop("legacy_no_c", None)
    with tempfile.TemporaryDirectory() as tmp:
        d.update(checkpoint_dir=os.path.join(tmp, "models"), self_ensemble=1, log_filename="")
        m = SuperResolution(_F(d))
        m.build_graph()
        m.init_all_variables()
        m.load_weights(dict(np.load(os.path.join(golden, "weights_L7_x2.npz"))))
        try:
            psnr = [m.do_for_evaluate(os.path.join(golden, "set5", f))[0] for f in g["files"]]
            crop = np.load(os.path.join(golden, "crop_L7_x2.npz"))
            out = m.do(crop["lr"], crop["bicubic"])
        finally:
            m.close()
    want = g["models"]["L7_x2"]["set5_psnr"]
    return {"set5_psnr_delta_db": float(max(abs(a - b) for a, b in zip(psnr, want))),
            "set5_psnr_mean_db": round(float(np.mean(psnr)), 4), "set5_psnr_mean_oracle_db": round(float(g["models"]["L7_x2"]["set5_mean"]), 4),
            "max_abs": float(np.max(np.abs(out - crop["output"]))),
            "bars": {"set5_psnr_delta_db": 1e-3, "max_abs": 1e-4},
            "model": "dcscn_L7_F32to8 x2 (c-DCSCN): the reference's shipped trained checkpoint -- the L12 / L8 blobs are not shipped "
                     "(.MISSING_LARGE_BLOBS), so the headline topology has no trained-weight parity; its synthetic-weight parity is tests/test_hip_parity.py",
            "against": "float64 oracle values committed under tests/golden (make_golden.py); README.md's published 37.15 dB is reproduced by them"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=PATCHES_PER_GPU, help="48x48 patches per GPU")
    ap.add_argument("--sub-batch-pixels", type=int, default=0, help="override the engine's pass size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ops", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--no-winograd", action="store_true", help="keep the 3x3 convs on the direct implicit-GEMM kernel")
    ap.add_argument("--cpu-sample", type=int, default=64, help="patches in the CPU baseline sample")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive timing of the host-buffer entry points")
    ap.add_argument("--no-opt-in", "--no-layer-by-layer", dest="no_extra_graph", action="store_true",
                    help="skip the extra timing of the layer-by-layer graph")
    ap.add_argument("--layer-by-layer", action="store_true",
                    help="headline on the reference's layers one by one (fold_linear_tail = 0) instead of the library default")
    ap.add_argument("--no-split16", action="store_true", help="headline on the pure f32 kernels (split16 = 0: conv_wino2 / conv_nin)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2 / C5 legs (other_configs) and the Set5 parity leg")
    ap.add_argument("--other-steps", type=int, default=20, help="timed steps of each other_configs leg")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip the extra strong-scaling leg (1024 patches in total)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --patches is the GLOBAL batch, split into contiguous shards over the ranks")
    ap.add_argument("--check-output", action="store_true",
                    help="after the timed steps, hash every output patch (sha256) and report the digest of the global batch "
                         "in patch order: equal for any number of ranks with --strong (tests/test_multi_rank_gpu.py)")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X; there is no CPU fallback for the HIP path")
    # test hook: DCSCN_BENCH_SHARE_GPU=1 lets several ranks share one device over gloo, to exercise the
    # multi-rank code path on a single-GPU box; real runs use one GPU per rank over RCCL
    share_gpu = os.environ.get("DCSCN_BENCH_SHARE_GPU") == "1"
    device_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))

    from dcscn_amd import engine
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dcscn_oracle as O          # synthetic-weight generator + CPU baseline only

    cfg = O.make_config(**MODEL_FLAGS)
    weights = O.synthetic_weights(cfg, seed=0)
    eng = engine.Engine(cfg, device=device_index)
    eng.load_weights(weights, winograd=False if args.no_winograd else None, fold_tail=False if args.layer_by_layer else None,
                     split16=False if (args.no_split16 or args.no_winograd) else None)
    if args.sub_batch_pixels:
        eng.set_option("sub_batch_pixels", args.sub_batch_pixels)
    folded = any("(folded)" in o["name"] for o in eng.ops())

    s = cfg["scale"]
    if args.strong:
        # the same global batch on every rank (seeded on the CPU so that it does not depend on the rank count),
        # contiguous shard [lo, hi) of it on this rank (dcscn-super-resolution_amd/shard.py)
        from dcscn_amd import shard
        lo, hi = shard.shard_bounds(args.patches, rank, world)
        n = hi - lo
        gcpu = torch.Generator(device="cpu")
        gcpu.manual_seed(1234)
        xg = torch.rand((args.patches, PATCH, PATCH, 1), generator=gcpu) * 255.0
        x2g = torch.rand((args.patches, PATCH * s, PATCH * s, 1), generator=gcpu) * 255.0
        x, x2 = xg[lo:hi].cuda(), x2g[lo:hi].cuda()
        del xg, x2g
    else:
        n = args.patches
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1234 + rank)
        x = torch.rand((n, PATCH, PATCH, 1), device="cuda", generator=gen) * 255.0
        x2 = torch.rand((n, PATCH * s, PATCH * s, 1), device="cuda", generator=gen) * 255.0
    y = torch.empty_like(x2)
    # a real (non-default) stream of our own: stream 0 / NULL would mean "the handle's own stream" to the C ABI
    tstream = torch.cuda.Stream()
    stream = tstream.cuda_stream
    torch.cuda.synchronize()          # inputs were produced on the default stream

    def run_forward():
        eng.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, PATCH, PATCH, stream)

    def step():
        # several ranks on ONE device (the single-GPU test rig, DCSCN_BENCH_SHARE_GPU=1) dispatch concurrently, like any other ranks:
        # the r03 cross-process corruption was v_pk_fma_f32 misbehaving beside another process's MFMAs (tools/xproc_triage.hip,
        # DESIGN.md section 6); the library is built without packed-f32 instructions since r04
        run_forward()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.set_option("profile", 1)      # HIP events around every launch, on the launch stream
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    per_op_ms = eng.profile(with_float32_plan=True)         # averaged over the timed steps
    f32_plan_ms = per_op_ms.pop()     # the gated float32 launches behind the pass (empty unless an image left the f16 range)
    eng.set_option("profile", 0)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if share_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if n and not bool(torch.isfinite(y).all().item()):
        raise SystemExit("non-finite output")
    digest = None
    if args.check_output:
        import hashlib
        yh = y.cpu().numpy()
        mine = [hashlib.sha256(yh[i].tobytes()).hexdigest() for i in range(n)]
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            mine = [d for p in parts for d in p]       # rank order == patch order (contiguous shards)
        digest = hashlib.sha256("".join(mine).encode()).hexdigest()
    global_patches = args.patches if args.strong else n * world
    # N > 1, weak headline: the strong-scaling leg of BASELINE configs[2] in the same line (one seeded global batch of
    # PATCHES_PER_GPU patches, contiguous shard per rank), timed exactly like the headline
    strong_leg = None
    if world > 1 and not args.strong and not args.no_strong_leg:
        from dcscn_amd import shard
        lo, hi = shard.shard_bounds(PATCHES_PER_GPU, rank, world)
        ns_ = hi - lo
        gcpu = torch.Generator(device="cpu")
        gcpu.manual_seed(1234)
        xg = torch.rand((PATCHES_PER_GPU, PATCH, PATCH, 1), generator=gcpu) * 255.0
        x2g = torch.rand((PATCHES_PER_GPU, PATCH * s, PATCH * s, 1), generator=gcpu) * 255.0
        xs_, x2s_ = xg[lo:hi].cuda(), x2g[lo:hi].cuda()
        del xg, x2g
        ys_ = torch.empty_like(x2s_)
        torch.cuda.synchronize()

        def sstep():
            eng.forward_device(xs_.data_ptr(), x2s_.data_ptr(), ys_.data_ptr(), ns_, PATCH, PATCH, stream)
        for _ in range(max(args.warmup, 1)):
            sstep()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            sstep()
        fence()
        el_s = time.perf_counter() - t1
        t = torch.tensor([el_s], device="cpu" if share_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_s = float(t.item())
        strong_leg = {"value": round(PATCHES_PER_GPU * PATCH * PATCH * args.steps / el_s / 1e6, 4), "unit": "LR Mpix/s",
                      "ms_per_step": round(el_s / args.steps * 1e3, 4), "scaling": "strong", "global_patches": PATCHES_PER_GPU,
                      "patches_on_rank0": ns_,
                      "note": "BASELINE configs[2] as written: 1024 patches in total, contiguous shards of 1024 / N per rank, "
                              "barrier + max over ranks like the headline"}

    if rank == 0:
        ops = eng.ops()
        lr_pixels = n * PATCH * PATCH                         # this rank's
        global_lr_pixels = global_patches * PATCH * PATCH
        # dominant kernel: the Winograd 3x3 launches (CNN2..12, B2, and Up-PS in the layer-by-layer graph)
        C3H = C3H_KERNELS                       # the split16 3x3 kernels: 4-wave workgroups / persistent 8-wave workgroups sharing the input tile
        dom = [(o, ms) for o, ms in zip(ops, per_op_ms) if o["kernel"] in ("conv_igemm", "conv_wino2") + C3H
               and o["kernel_size"] == 3 and o["out_channels"] > 1]
        dom_kernels = sorted({o["kernel"] for o, _ in dom})
        on_f16 = any(k in C3H for k in dom_kernels)
        DOM_PREFIX[0] = "conv3_h" if on_f16 else "conv_wino"
        dom_peak = PEAK_F16_MFMA_TFLOPS if on_f16 else PEAK_F32_MFMA_TFLOPS
        # FLOPs the instruction stream would issue without channel padding: 3 f16 products per MAC (conv3_h), or the 16/36
        # of F(2x2,3x3) (conv_wino2)
        dom_useful = dom_flop_useful = sum(2.0 * o["macs_per_lr_pixel"] * (3.0 if o["kernel"] in C3H else 16.0 / 36.0 if o["kernel"] == "conv_wino2" else 1.0)
                                            for o, _ in dom) * lr_pixels
        dom_flop = sum(2.0 * o["macs_per_lr_pixel"] for o, _ in dom) * lr_pixels
        dom_bytes = sum(float(o["bytes_per_lr_pixel"]) for o, _ in dom) * lr_pixels
        dom_ms = sum(ms for _, ms in dom)
        algorithmic = dom_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        dom_exec = sum(2.0 * o["executed_macs_per_lr_pixel"] for o, _ in dom) * lr_pixels
        executed = dom_exec / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        total_macs = sum(o["macs_per_lr_pixel"] for o in ops)
        kernel_ms = sum(per_op_ms)
        per_kernel = {}
        for o, ms in zip(ops, per_op_ms):
            key = o["kernel"] + ("_%dx%d" % (o["kernel_size"], o["kernel_size"]) if o["kernel"] in ("conv_igemm", "conv_wino2") + C3H else "")
            per_kernel[key] = per_kernel.get(key, 0.0) + ms
        if args.ops:
            for o, ms in zip(ops, per_op_ms):
                fl = 2.0 * o["macs_per_lr_pixel"] * lr_pixels
                fx = 2.0 * o["executed_macs_per_lr_pixel"] * lr_pixels
                by = o["bytes_per_lr_pixel"] * lr_pixels
                print("%-22s %-11s k%d %4d->%-4d res%d mt%d nt%-2d kc%-2d tiles%d  %8.3f ms  %7.2f TFLOP/s (alg)  %7.2f TFLOP/s (exec)  %7.1f GB/s"
                      % (o["name"], o["kernel"], o["kernel_size"], o["in_channels"], o["out_channels"], o["resolution"],
                         o["mt"], o["nt"], o["kc"], o["n_tiles"], ms, fl / (ms * 1e-3) / 1e12 if ms else 0,
                         fx / (ms * 1e-3) / 1e12 if ms else 0, by / (ms * 1e-3) / 1e9 if ms else 0), file=sys.stderr)
        nin_ms = sum(ms for o, ms in zip(ops, per_op_ms) if o["kernel"] in ("conv_igemm", "conv_nin", "conv_nin_h") and o["kernel_size"] == 1)
        full_workload = n == PATCHES_PER_GPU and not args.no_winograd
        traffic, north_star = pmc_replay(dom_ms, nin_ms, dom_bytes, {o["kernel"] for o in ops}) if full_workload else (None, None)
        graph = ("linear tail (Up-PS conv + depth_to_space + R-CNN1) folded into one 5x5 conv -- library default, include/dcscn.h fold_linear_tail"
                 if folded else "the reference's layers, one launch per layer (B1+A1 share a launch)")
        result = {
            "metric": "LR Mpixels/sec at 48x48 patches, L12_F196to48 x2",
            "value": round(global_lr_pixels * args.steps / elapsed / 1e6, 4),
            "unit": "LR Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": ("f32-equivalent (3x3 stack, wide 1x1 convs and folded tail: f16 hi/lo split, 3 products per MAC, f32 accumulate -- "
                      "error at the f32 kernels' level, same parity bars; everything else f32)") if on_f16 else "f32",
            "data": "synthetic (uniform 0-255 Y patches, seeded He-init weights; trained L12 blobs are not shipped)",
            "config": {
                "workload": "%s x2 forward, %s (BASELINE.json configs[2])" % (
                    MODEL_NAME, ("%d 48x48 Y patches in total, %d on rank 0" % (global_patches, n)) if args.strong
                    else "%d 48x48 Y patches per GPU" % n),
                "patches_per_gpu": n,
                "global_patches": global_patches,
                "parallelism": "image-shard x%d, no collective" % world,
                "flop_per_lr_pixel_direct_form": 2 * total_macs,
                "graph": graph,
                "arithmetic": "split16, tensors between its launches pre-split (p16; include/dcscn.h)" if on_f16 else "f32 kernels (split16 = 0)",
            },
            "roofline": {
                "kernel": "%s 3x3 (%s), %d launches/pass" % ("+".join(dom_kernels), "v_mfma_f32_16x16x32_f16, f16 hi/lo x 3 products" if on_f16
                                                             else "v_mfma_f32_16x16x4_f32", len(dom)),
                "bound": "mfma",
                "achieved": round(executed, 3),
                "peak": dom_peak,
                "unit": "TFLOP/s",
                "frac": round(executed / dom_peak, 4),
                "useful_frac": round(dom_useful / (dom_ms * 1e-3) / 1e12 / dom_peak, 4) if dom_ms > 0 else None,
                "algorithmic_frac": round(algorithmic / dom_peak, 4),
                "frac_of_sustained": round(executed / SUSTAINED_F16_MFMA_TFLOPS, 4) if on_f16 else None,
                "traffic": traffic["bytes_per_step"] if traffic else None,
                "traffic_detail": traffic if traffic else ({"traffic": None, "reason": north_star["reason"]}
                                                           if north_star and "reason" in north_star else None),
                "executed_flop_per_step": dom_exec,
                "algorithmic_flop_per_step": dom_flop,
                "vs_f32_peak": {"algorithmic_tflops": round(algorithmic, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                                "ratio": round(algorithmic / PEAK_F32_MFMA_TFLOPS, 4),
                                "note": "direct-form f32 FLOPs of SURVEY.md 8(d) for the same launches per second, against the f32 "
                                        "MFMA / VALU peak the reference's arithmetic is bound by"},
                "note": ("achieved = f16 FLOPs the kernel issues (3 products per MAC, input channels padded to 32, output channels to 16) "
                         "per second of its own launch time (HIP events on the launch stream); frac = that against the dense f16 peak "
                         "(2500, nominal 2.4 GHz); useful_frac = without the channel padding (3 x the algorithmic FLOPs: what f32-accurate "
                         "arithmetic on this pipe has to issue); algorithmic_frac = the reference's own arithmetic (SURVEY 8(d) FLOPs, 1 x) "
                         "against the same peak.  frac_of_sustained = against 2.1 PFLOP/s, this repository's bare-MFMA microbenchmark with "
                         "constant operands (profiles/r03_mfma_two_waves.txt); MI355X_MICROARCH.md records 2.38-2.50 PFLOP/s for dense "
                         "bf16 / f16, and with random operands in conv3_h's register pattern the same microbenchmark sustains 9.4 ns "
                         "per MFMA = 1.78 PFLOP/s (profiles/r04_mfma_operands.txt)." if on_f16 else
                         "achieved = FLOPs the kernel issues (Winograd F(2x2,3x3): 16/36 of the direct form, plus channel padding to 16 / 8) "
                         "per second of its own launch time (HIP events on the launch stream); frac = matrix-pipe utilisation."),
                "kernel_ms_per_step": round(dom_ms, 4),
            },
            "kernel_ms_per_step": dict({k: round(v, 4) for k, v in sorted(per_kernel.items())}, float32_plan_gated=round(f32_plan_ms, 4)),
            "whole_net_tflops_direct_form_equivalent": round(2.0 * total_macs * lr_pixels / (kernel_ms * 1e-3) / 1e12, 3) if kernel_ms else None,
        }
        if strong_leg is not None:
            result["strong_scaling"] = strong_leg
        if digest is not None:
            result["output_sha256"] = digest
        if north_star:
            result["north_star"] = north_star
        if world == 1 and not args.no_cpu_baseline:
            import cpu_path_torch as T
            cs = min(args.cpu_sample, n)
            xs = x[:cs].cpu().numpy()
            x2s = x2[:cs].cpu().numpy()
            sec, threads, ycpu, layout = T.time_cpu_path(cfg, weights, xs, x2s, reps=2)
            dev_err = float(np.max(np.abs(ycpu - y[:cs].cpu().numpy())))
            result["cpu_baseline"] = {
                "value": round(cs * PATCH * PATCH / sec / 1e6, 5),
                "unit": "LR Mpix/s",
                "cores": threads,
                "kind": "port",
                "layout": layout,
                "achieved_gflops": round(sum(o["macs_per_lr_pixel"] for o in ops) * 2.0 * cs * PATCH * PATCH / sec / 1e9, 1),
                "host_cores": os.cpu_count(),
                "sample": "%d of the %d patches, float32 torch-CPU (oneDNN; NCHW and channels-last both timed at all host cores and "
                          "at a quarter of them, the fastest kept: %s, %d threads) restatement of the reference graph (TensorFlow "
                          "not installable), best of 2 after 1 warm-up per setting, %.2f s/forward" % (cs, n, layout, threads, sec),
                "max_abs_diff_vs_hip": dev_err,
            }
        if world == 1 and not args.no_host_path:
            # PCIe-inclusive rate of the host-buffer entry points (never `value`): numpy in, numpy out, synchronous
            try:
                # numpy-owned copies: buffers handed out by torch's CPU allocator upload 20x slower through hipMemcpy
                xh, x2h = x.cpu().numpy().copy(), x2.cpu().numpy().copy()
                yh = np.empty_like(x2h)                   # result buffer reused across calls (no page faults under the download)
                eng.forward(xh, x2h, out=yh)
                t1 = time.perf_counter()
                for _ in range(3):
                    eng.forward(xh, x2h, out=yh)
                th = (time.perf_counter() - t1) / 3
                eng.forward_lr(xh, out=yh)
                t1 = time.perf_counter()
                for _ in range(3):
                    eng.forward_lr(xh, out=yh)
                tl = (time.perf_counter() - t1) / 3
                result["host_path"] = {
                    "dcscn_forward": {"value": round(lr_pixels / th / 1e6, 4), "ms_per_step": round(th * 1e3, 3),
                                      "note": "x and x2 uploaded, y downloaded (%.1f MB over PCIe per step), in 4 chunks overlapped with the kernels" % ((xh.nbytes + 2 * x2h.nbytes) / 1e6)},
                    "dcscn_forward_lr": {"value": round(lr_pixels / tl / 1e6, 4), "ms_per_step": round(tl * 1e3, 3),
                                         "note": "x uploaded, x2 = bicubic(x) built on the device, y downloaded; not comparable to "
                                                 "`value` input-wise (x2 here is the real bicubic, not noise) but the same work"},
                    "unit": "LR Mpix/s",
                }
            except Exception as exc:
                result["host_path"] = {"error": str(exc)}
        if world == 1 and not args.no_extra_graph:
            # beside the headline: the same step replayed from a hipGraph (option graph_replay: captured on the second call with
            # the same arguments) -- one graph launch instead of the pass's kernel launches
            try:
                y_plain = y.clone()
                eng.set_option("graph_replay", 1)
                for _ in range(max(args.warmup, 3)):
                    step()
                eng.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                eng.synchronize()
                el4 = time.perf_counter() - t1
                result["graph_replay"] = {
                    "value": round(lr_pixels * args.steps / el4 / 1e6, 4), "unit": "LR Mpix/s",
                    "ms_per_step": round(el4 / args.steps * 1e3, 4),
                    "bit_identical_to_plain_launches": bool(torch.equal(y, y_plain)),
                    "note": "dcscn_set_option(graph_replay, 1): opt-in (the pointers must repeat); not the headline",
                }
                eng.set_option("graph_replay", 0)
            except Exception as exc:
                result["graph_replay"] = {"error": str(exc)}
        if world == 1 and on_f16 and not args.no_extra_graph:
            # beside the headline: the same engine with float32 tensors between the split16 launches (p16 = 0: the r04 data path)
            try:
                y_p16 = y.clone()
                n_p16 = eng.num_presplit_tensors()
                eng.set_option("p16", 0)
                for _ in range(max(args.warmup, 1)):
                    step()
                eng.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                eng.synchronize()
                el5 = time.perf_counter() - t1
                result["float32_tensors"] = {
                    "value": round(lr_pixels * args.steps / el5 / 1e6, 4), "unit": "LR Mpix/s",
                    "ms_per_step": round(el5 / args.steps * 1e3, 4),
                    "bit_identical_to_headline": bool(torch.equal(y, y_p16)),
                    "p16_tensors_in_headline": n_p16,
                    "note": "dcscn_set_option(p16, 0): float32 NHWC tensors between the split16 launches, split in every consumer (r04); the headline "
                            "keeps them pre-split (csrc/p16.hpp): same products in the same order",
                }
                eng.set_option("p16", 1)
                step()
                eng.synchronize()
            except Exception as exc:
                result["float32_tensors"] = {"error": str(exc)}
        if world == 1 and on_f16 and not args.no_extra_graph:
            # beside the headline: the same engine on the pure f32 kernels (split16 = 0), same inputs, timed the same way
            try:
                y_h16 = y.clone()
                eng.set_option("split16", 0)
                for _ in range(max(args.warmup, 1)):
                    step()
                eng.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                eng.synchronize()
                el3 = time.perf_counter() - t1
                result["f32_path"] = {
                    "value": round(lr_pixels * args.steps / el3 / 1e6, 4), "unit": "LR Mpix/s",
                    "ms_per_step": round(el3 / args.steps * 1e3, 4),
                    "max_abs_diff_vs_split16": float((y - y_h16).abs().max().item()),
                    "note": "split16 = 0: 3x3 convs on conv_wino2 (f32 Winograd on v_mfma_f32_16x16x4_f32), 1x1 on conv_nin -- the r02 headline path",
                }
                eng.set_option("split16", 1)
                step()
                eng.synchronize()
            except Exception as exc:
                result["f32_path"] = {"error": str(exc)}
        if world == 1 and folded and not args.no_extra_graph:
            # beside the headline: the same inputs through the layer-by-layer graph (fold_linear_tail = 0), timed the same way
            try:
                y_ref = y.clone()
                eng2 = engine.Engine(cfg, device=device_index)
                eng2.load_weights(weights, winograd=False if args.no_winograd else None, fold_tail=False)
                if args.sub_batch_pixels:
                    eng2.set_option("sub_batch_pixels", args.sub_batch_pixels)
                for _ in range(max(args.warmup, 1)):
                    eng2.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, PATCH, PATCH, stream)
                eng2.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    eng2.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, PATCH, PATCH, stream)
                eng2.synchronize()
                el2 = time.perf_counter() - t1
                result["layer_by_layer"] = {
                    "value": round(lr_pixels * args.steps / el2 / 1e6, 4), "unit": "LR Mpix/s",
                    "ms_per_step": round(el2 / args.steps * 1e3, 4),
                    "max_abs_diff_vs_default_graph": float((y - y_ref).abs().max().item()),
                    "note": "fold_linear_tail = 0: Up-PS conv, depth_to_space and R-CNN1 as separate launches (the reference's graph)",
                }
                eng2.close()
            except Exception as exc:      # the headline line must survive a failure of the extra leg
                result["layer_by_layer"] = {"error": str(exc)}
        if world == 1 and not args.no_other_configs and not args.no_extra_graph:
            # SURVEY.md 8(d) "also C2, C5": the other single-GPU BASELINE configurations, and the metric's "PSNR delta vs ref on Set5"
            result["other_configs"] = {}
            for key in OTHER_CONFIGS:
                try:
                    result["other_configs"][key] = time_other_config(engine, O, torch, key, args.other_steps, args.warmup, device_index, stream)
                except Exception as exc:
                    result["other_configs"][key] = {"error": str(exc)}
            try:
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):      # (the model prints like the reference's: stdout carries the ONE JSON line only)
                    result["parity"] = parity_leg()
            except Exception as exc:
                result["parity"] = {"error": str(exc)}
        print(json.dumps(result), flush=True)

    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
