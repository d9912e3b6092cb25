"""float32 torch-CPU (oneDNN) restatement of the DCSCN graph -- TEST / BASELINE INFRASTRUCTURE ONLY.

This is the timed "reference CPU path" stand-in of bench.py's ``cpu_baseline`` leg (kind "port"):
TensorFlow, which the reference runs on (DCSCN.py:565-578), cannot be installed here, so the same
graph (``dcscn_oracle.build_topology``, restating DCSCN.py:222-325) is executed with torch's CPU
convolutions on all host cores.  It is also used by tests as a float32 cross-check of the float64
numpy oracle.  The shipped package never imports this file.
"""

import numpy as np
import torch
import torch.nn.functional as F

import dcscn_oracle as O


class TorchCpuModel:
    def __init__(self, cfg, weights, dtype=torch.float32, channels_last=False):
        self.cfg = cfg
        self.channels_last = channels_last
        self.ops = O.build_topology(cfg)
        self.dtype = dtype
        self.w = {}
        for name, arr in weights.items():
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype)
            leaf = name.rsplit("/", 1)[-1]
            if leaf == "Tconv_W":
                t = t.permute(3, 2, 0, 1).contiguous()            # [kh,kw,out,in] -> [in,out,kh,kw]
            elif leaf in ("conv_W", "pointwise_W"):
                t = t.permute(3, 2, 0, 1).contiguous()            # HWIO -> OIHW
            elif leaf == "depthwise_W":
                t = t.permute(2, 3, 0, 1).contiguous()            # [k,k,C,1] -> [C,1,k,k]
            if channels_last and t.dim() == 4:
                t = t.contiguous(memory_format=torch.channels_last)           # oneDNN's NHWC kernels
            self.w[name] = t

    @torch.no_grad()
    def forward(self, x, x2):
        """x: [N,H,W,1], x2: [N,sH,sW,1] numpy -> [N,sH,sW,1] numpy."""
        t = {"x": torch.from_numpy(np.ascontiguousarray(x)).to(self.dtype).permute(0, 3, 1, 2),
             "x2": torch.from_numpy(np.ascontiguousarray(x2)).to(self.dtype).permute(0, 3, 1, 2)}
        if self.channels_last:
            t = {k: v.contiguous(memory_format=torch.channels_last) for k, v in t.items()}
        for op in self.ops:
            kind = op["op"]
            if kind == "conv":
                v, k = op["var"], op["k"]
                h = t[op["src"]]
                if op["ds"]:
                    h = F.conv2d(h, self.w[v + "/depthwise_W"], padding=k // 2, groups=h.shape[1])
                    h = F.conv2d(h, self.w[v + "/pointwise_W"])
                else:
                    h = F.conv2d(h, self.w[v + "/conv_W"], padding=k // 2)
                if op["bias"]:
                    h = h + self.w[v + "/conv_B"].view(1, -1, 1, 1)
                act = op["act"]
                if act == "prelu":
                    a = self.w[v + "/prelu/" + op["name"] + "_prelu"].view(1, -1, 1, 1)
                    h = torch.relu(h) + a * (h - torch.abs(h)) * 0.5            # tf_graph.py:94
                elif act == "relu":
                    h = torch.relu(h)
                elif act == "leaky_relu":
                    h = torch.maximum(h, 0.1 * h)
                elif act == "sigmoid":
                    h = torch.sigmoid(h)
                elif act == "tanh":
                    h = torch.tanh(h)
                elif act == "selu":
                    h = F.selu(h)
                elif act not in (None, ""):
                    raise NameError("Not implemented activator:%s" % act)
                t[op["dst"]] = h
            elif kind == "concat":
                t[op["dst"]] = torch.cat([t[s] for s in op["srcs"]], dim=1)
            elif kind == "depth_to_space":
                b = op["block"]
                h = t[op["src"]]
                n, c, hh, ww = h.shape
                co = c // (b * b)
                # TF channel order: (i*b + j)*co + c  ->  [n, co, hh*b, ww*b]
                h = h.view(n, b, b, co, hh, ww).permute(0, 3, 4, 1, 5, 2).reshape(n, co, hh * b, ww * b)
                t[op["dst"]] = h
            elif kind == "conv_transpose":
                sc = op["scale"]
                w = self.w[op["var"] + "/Tconv_W"]
                t[op["dst"]] = F.conv_transpose2d(t[op["src"]], w, stride=sc, padding=(w.shape[2] - sc) // 2)
            elif kind == "add":
                t[op["dst"]] = t[op["srcs"][0]] + t[op["srcs"][1]]
        return t["y_"].permute(0, 2, 3, 1).contiguous().numpy()


def time_cpu_path(cfg, weights, x, x2, reps=3, threads=None):
    """Times the float32 CPU forward.  ``threads=None`` tries all host cores and a quarter of them
    (oneDNN convolutions on small patches stop scaling long before 256 threads), in NCHW and in channels-last
    layout, and keeps the fastest.  Returns (seconds per forward (best), threads used, output, layout)."""
    import os
    import time
    cores = os.cpu_count() or 1
    candidates = [threads] if threads else sorted({cores, max(1, cores // 4)}, reverse=True)
    best, best_threads, best_layout, y = float("inf"), candidates[0], "NCHW", None
    for layout in ("NCHW", "channels_last"):
        model = TorchCpuModel(cfg, weights, channels_last=layout == "channels_last")
        for th in candidates:
            torch.set_num_threads(th)
            out = model.forward(x, x2)          # warm-up
            if y is None:
                y = out
            for _ in range(reps):
                t0 = time.perf_counter()
                out = model.forward(x, x2)
                dt = time.perf_counter() - t0
                if dt < best:
                    best, best_threads, best_layout, y = dt, th, layout, out
    return best, best_threads, y, best_layout
