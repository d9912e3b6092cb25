"""The oracle's graph against what the REFERENCE itself ships (tests/golden/ref_graphs.json, generated from
/root/reference/models by tests/golden/make_ref_graphs.py): the variable names and shapes of all twelve checkpoints
-- including the six L12 / L8 models the headline metric is quoted on, of which the reference ships only the .index --
and, for the six models with a MetaGraphDef, the wiring of the inference subgraph: which conv reads which tensor,
bias adds, the PReLU subgraph, the ORDER of the concat inputs, DepthToSpace block sizes and the final add.

This pins oracle.variable_shapes / oracle.build_topology (and, on the GPU box, the library's own tensor list) to
reference-held data for every BASELINE config, not only to the PSNR table of the L7 / L2 models."""
import json
import os

import pytest

from conftest import GOLDEN

L7 = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8, reconstruct_layers=0,
          pixel_shuffler_filters=1)
L2 = dict(layers=2, filters=4, min_filters=4, use_nin=False, reconstruct_filters=4, legacy_no_c=True)
# checkpoint name -> flags (README.md:80,86 of the reference)
MODELS = {
    "dcscn_L12_F196to48_NIN_A64_PS_R1F32": dict(),
    "dcscn_L12_F196to48_Sc3_NIN_A64_PS_R1F32": dict(scale=3),
    "dcscn_L12_F196to48_Sc4_NIN_A64_PS_R1F32": dict(scale=4),
    "dcscn_L8_F96to48_NIN_A64_PS_R1F32": dict(layers=8, filters=96),
    "dcscn_L8_F96to48_Sc3_NIN_A64_PS_R1F32": dict(layers=8, filters=96, scale=3),
    "dcscn_L8_F96to48_Sc4_NIN_A64_PS_R1F32": dict(layers=8, filters=96, scale=4),
    "dcscn_L7_F32to8_G1.20_NIN_A24_B8_PS_R1F32": dict(L7),
    "dcscn_L7_F32to8_G1.20_Sc3_NIN_A24_B8_PS_R1F32": dict(L7, scale=3),
    "dcscn_L7_F32to8_G1.20_Sc4_NIN_A24_B8_PS_R1F32": dict(L7, scale=4),
    "dcscn_L7_F32to8_G1.20_Sc4_NIN_A24_B8_PS_DS_R1F32": dict(L7, scale=4, depthwise_separable=True),
    "dcscn_L2_F4to4_PS_R1F4": dict(L2),
    "dcscn_L2_F4to4_Sc4_PS_R1F4": dict(L2, scale=4),
}


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(GOLDEN, "ref_graphs.json")) as f:
        return json.load(f)["models"]


def test_every_shipped_checkpoint_is_covered(ref):
    assert sorted(ref) == sorted(MODELS)
    assert sum(1 for m in ref.values() if "graph" in m) == 6          # the L7 / L2 models carry a MetaGraphDef
    assert sorted(n for n, m in ref.items() if not m["has_data"]) == sorted(n for n in MODELS if "_L12_" in n or "_L8_" in n)


@pytest.mark.parametrize("name", sorted(MODELS))
def test_variable_names_and_shapes_match_the_reference_index(oracle, ref, name):
    cfg = oracle.make_config(**MODELS[name])
    want = {k: tuple(v) for k, v in ref[name]["variables"].items()}
    got = {k: tuple(v) for k, v in oracle.variable_shapes(cfg).items()}
    if cfg["depthwise_separable"]:
        # build_depthwise_separable_conv also creates a conv_W it never uses (tf_graph.py:166-168): the checkpoint holds it
        extra = set(want) - set(got)
        assert extra and all(k.endswith("/conv_W") for k in extra), sorted(extra)
        want = {k: v for k, v in want.items() if k not in extra}
    assert got == want


def test_parameter_counts_of_the_bench_models(oracle, ref):
    """1,754,942 parameters for the L12 x2 model (the 'Complexity' line of the reference log)."""
    import numpy as np
    counts = {n: sum(int(np.prod(s)) for s in ref[n]["variables"].values()) for n in ref}
    assert counts["dcscn_L12_F196to48_NIN_A64_PS_R1F32"] == 1754942
    assert counts["dcscn_L8_F96to48_NIN_A64_PS_R1F32"] == 687268
    assert counts["dcscn_L12_F196to48_Sc4_NIN_A64_PS_R1F32"] == 2087102


def _tensor_name(ref_name):
    """reference tensor producer -> the oracle's tensor name"""
    if ref_name == "Concat/H_concat":
        return "H_concat"
    if ref_name.endswith("/DepthToSpace"):
        return ref_name.split("/")[0]
    return ref_name


@pytest.mark.parametrize("name", sorted(n for n in MODELS if "_L7_" in n or "_L2_" in n))
def test_topology_matches_the_reference_metagraph(oracle, ref, name):
    cfg = oracle.make_config(**MODELS[name])
    g = ref[name]["graph"]
    topo = oracle.build_topology(cfg)
    convs = [op for op in topo if op["op"] == "conv"]
    assert [c["var"] for c in convs] == [layer["scope"] for layer in g["layers"]]          # build order
    by_dst = {}
    for op in topo:
        if op["op"] == "conv":
            by_dst[op["dst"]] = op["var"]                                                     # a conv's output is named by its scope
    for c, layer in zip(convs, g["layers"]):
        v = c["var"]
        src = by_dst.get(c["src"], c["src"])
        assert src == _tensor_name(layer["src"]), (v, src, layer["src"])
        if c["ds"]:
            assert layer["ops"] == ["DepthwiseConv2dNative", "Conv2D"]
            assert layer["filters"] == [v + "/depthwise_W", v + "/pointwise_W"]
        else:
            assert layer["ops"] == ["Conv2D"] and layer["filters"] == [v + "/conv_W"]
        assert layer["strides"] == [1, 1, 1, 1] and layer["padding"] == "SAME" and layer["data_format"] == "NHWC"
        assert layer["bias"] == (v + "/conv_B" if c["bias"] else None)
        if c["act"] == "prelu":
            a = layer["activator"]
            # relu(x) + alpha * (x - |x|) * 0.5  (tf_graph.py:89-94)
            assert a["kind"] == "prelu" and a["alpha"] == v + "/prelu/" + c["name"] + "_prelu"
            assert a["ops"] == ["Abs", "Add", "Mul", "Mul", "Relu", "Sub"] and a["half_const"] == 0.5
        else:
            assert c["act"] is None and layer["activator"] is None                            # Up-PS*_CNN and the last R-CNN
    want_concats = [(op["dst"], [by_dst.get(s, s) for s in op["srcs"]]) for op in topo if op["op"] == "concat"]
    got_concats = [(_tensor_name(c["name"]), [_tensor_name(s) for s in c["srcs"]]) for c in g["concats"]]
    assert got_concats == want_concats                                                        # Concat2 = [B2, A1]
    want_d2s = [(op["dst"], by_dst.get(op["src"], op["src"]), op["block"]) for op in topo if op["op"] == "depth_to_space"]
    got_d2s = [(_tensor_name(d["name"]), d["src"], d["block_size"]) for d in g["depth_to_space"]]
    assert got_d2s == want_d2s
    add = [op for op in topo if op["op"] == "add"]
    assert len(add) == 1 and [by_dst.get(s, s) for s in add[0]["srcs"]] == g["output"]["srcs"] and g["output"]["name"] in ("output", "add")   # the x3 model was saved by an older revision without name="output"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_library_tensor_list_matches_the_reference_index(oracle, ref, name):
    """dcscn_tensor_info (the names dcscn_set_tensor accepts) == the inference variables of the reference checkpoint."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**MODELS[name])
    want = {k: tuple(v) for k, v in ref[name]["variables"].items()}
    with engine.Engine(cfg, device=0) as eng:
        got = {n: tuple(s) for n, s in eng.tensor_specs()}
    if cfg["depthwise_separable"]:
        want = {k: v for k, v in want.items() if not (k.endswith("/conv_W") and k not in got)}
    assert got == want
