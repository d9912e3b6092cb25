"""ctypes wrapper of oracle/libdcscn_ref.so (plain-C restatement, see dcscn_ref.c) -- TEST INFRASTRUCTURE.
``forward`` walks the same topology as dcscn_oracle.forward but executes every op in C."""
import ctypes
import os
import subprocess

import numpy as np

import dcscn_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_ACT = {None: 0, "": 0, "prelu": 1, "relu": 2, "leaky_relu": 3, "sigmoid": 4, "tanh": 5, "selu": 6}


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libdcscn_ref.so")
        if not os.path.isfile(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "dcscn_ref.c")):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = ctypes.CDLL(path)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def forward(cfg, weights, x, x2):
    L = lib()
    t = {"x": np.ascontiguousarray(x, np.float64), "x2": np.ascontiguousarray(x2, np.float64)}
    for op in O.build_topology(cfg):
        kind = op["op"]
        if kind == "conv":
            v, k = op["var"], op["k"]
            src = t[op["src"]]
            n, h, w, cin = src.shape
            if op["ds"]:
                dw = np.ascontiguousarray(weights[v + "/depthwise_W"], np.float64)
                mid = np.empty_like(src)
                L.ref_depthwise_same(_p(src), _p(dw), _p(mid), n, h, w, cin, k)
                src, k = mid, 1
                wt = np.ascontiguousarray(weights[v + "/pointwise_W"], np.float64)
            else:
                wt = np.ascontiguousarray(weights[v + "/conv_W"], np.float64)
            cout = wt.shape[3]
            out = np.empty((n, h, w, cout), np.float64)
            L.ref_conv2d_same(_p(src), _p(wt), _p(out), n, h, w, cin, cout, k)
            bias = np.ascontiguousarray(weights[v + "/conv_B"], np.float64) if op["bias"] else None
            alpha = np.ascontiguousarray(weights[v + "/prelu/" + op["name"] + "_prelu"], np.float64) \
                if op["act"] == "prelu" else None
            if bias is not None or op["act"]:
                L.ref_bias_act(_p(out), _p(bias) if bias is not None else None, _p(alpha) if alpha is not None else None,
                               ctypes.c_size_t(n * h * w), cout, _ACT[op["act"]])
            t[op["dst"]] = out
        elif kind == "concat":
            t[op["dst"]] = np.ascontiguousarray(np.concatenate([t[s] for s in op["srcs"]], axis=3))
        elif kind == "depth_to_space":
            src = t[op["src"]]
            n, h, w, c = src.shape
            b = op["block"]
            out = np.empty((n, h * b, w * b, c // (b * b)), np.float64)
            L.ref_depth_to_space(_p(src), _p(out), n, h, w, c, b)
            t[op["dst"]] = out
        elif kind == "add":
            t[op["dst"]] = t[op["srcs"][0]] + t[op["srcs"][1]]
    return t["y_"]
