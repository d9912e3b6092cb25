"""Flag surface of the reference (helper/args.py:13-107): same names, types and defaults, hosted on
the absl-like registry of ``dcscn-super-resolution_amd/flags.py`` instead of ``tf.app.flags``.

Usage is unchanged:  ``from helper import args``; ``args.flags.DEFINE_string(...)``;
``FLAGS = args.get()``; and ``args.run(main)`` where the reference calls ``tf.app.run()``.
Training-only flags are accepted (so existing command lines keep parsing) but nothing reads them.
"""

import sys

import numpy as np

import dcscn_amd
from dcscn_amd import flags as flags          # noqa: F401  (the reference exposes args.flags)

FLAGS = flags.FLAGS
run = flags.run

_I, _F, _B, _S = flags.DEFINE_integer, flags.DEFINE_float, flags.DEFINE_boolean, flags.DEFINE_string

_TABLE = [
    # ---- model (args.py:17-36)
    (_I, "scale", 2, "super-resolution factor: 2, 3 or 4"),
    (_I, "layers", 12, "feature-extraction conv layers"),
    (_I, "filters", 196, "filters of the first feature layer"),
    (_I, "min_filters", 48, "filters of the last feature layer"),
    (_F, "filters_decay_gamma", 1.5, "decay exponent of the filter count from filters to min_filters"),
    (_B, "use_nin", True, "network-in-network reconstruction (A1 / B1 / B2)"),
    (_I, "nin_filters", 64, "filters of A1"),
    (_I, "nin_filters2", 32, "filters of B1 and B2"),
    (_I, "cnn_size", 3, "conv kernel size"),
    (_I, "reconstruct_layers", 1, "reconstruction conv layers (0 behaves as 1)"),
    (_I, "reconstruct_filters", 32, "filters of the extra reconstruction layers"),
    (_F, "dropout_rate", 0.8, "keep probability while training; inference always keeps everything"),
    (_S, "activator", "prelu", "relu, leaky_relu, prelu, sigmoid, tanh or selu"),
    (_B, "pixel_shuffler", True, "pixel-shuffler upsampling; false = transposed-conv upsampler (Up-TCNN, tf_graph.py:219-236)"),
    (_I, "pixel_shuffler_filters", 0, "pixel-shuffler output channels, 0 = as many as its input"),
    (_I, "self_ensemble", 8, "flipped / rotated copies averaged per image, 1 to 8"),
    (_B, "batch_norm", False, "batch normalisation (not implemented)"),
    (_B, "depthwise_separable", False, "depthwise-separable convs in place of every conv"),
    # ---- training (args.py:39-59), parsed for command-line compatibility only
    (_B, "bicubic_init", True, "training only"),
    (_F, "clipping_norm", 5, "training only"),
    (_S, "initializer", "he", "weight initialiser used before a checkpoint is loaded"),
    (_F, "weight_dev", 0.01, "training only"),
    (_F, "l2_decay", 0.0001, "training only"),
    (_S, "optimizer", "adam", "training only"),
    (_F, "beta1", 0.9, "training only"),
    (_F, "beta2", 0.999, "training only"),
    (_F, "epsilon", 1e-8, "training only"),
    (_F, "momentum", 0.9, "training only"),
    (_I, "batch_num", 20, "training only"),
    (_I, "batch_image_size", 48, "training only"),
    (_I, "stride_size", 0, "training only"),
    (_I, "training_images", 24000, "training only"),
    (_B, "use_l1_loss", False, "training only"),
    (_F, "initial_lr", 0.002, "training only"),
    (_F, "lr_decay", 0.5, "training only"),
    (_I, "lr_decay_epoch", 9, "training only"),
    (_F, "end_lr", 2e-5, "training only"),
    # ---- datasets (args.py:62-65)
    (_S, "dataset", "bsd200", "training only"),
    (_S, "test_dataset", "set5", "directory under data_dir to evaluate: set5, set14, bsd100, ... or all"),
    (_I, "tests", 1, "number of trained models (trials) to evaluate"),
    (_B, "do_benchmark", False, "training only"),
    # ---- image processing (args.py:68-74)
    (_F, "max_value", 255, "pixel value range the network works in"),
    (_I, "channels", 1, "image channels fed to the network; only 1 (Y of YCbCr) is supported"),
    (_I, "psnr_calc_border_size", -1, "border shaved before PSNR; negative means the scale factor"),
    (_B, "build_batch", False, "training only"),
    # ---- environment (args.py:77-85)
    (_S, "checkpoint_dir", "models", "directory of the TF checkpoints"),
    (_S, "graph_dir", "graphs", "unused"),
    (_S, "data_dir", "data", "directory of the image datasets"),
    (_S, "batch_dir", "batch_data", "training only"),
    (_S, "output_dir", "output", "directory result images are written to"),
    (_S, "tf_log_dir", "tf_log", "unused (no TensorBoard)"),
    (_S, "log_filename", "log.txt", "log file"),
    (_S, "model_name", "", "explicit model name instead of the one derived from the flags"),
    (_S, "load_model_name", "", "checkpoint to load: a file stem or 'default'"),
    # ---- logging / device (args.py:88-94)
    (_B, "initialize_tf_log", True, "unused"),
    (_B, "enable_log", True, "unused"),
    (_B, "save_weights", True, "unused"),
    (_B, "save_images", False, "unused"),
    (_I, "save_images_num", 20, "unused"),
    (_B, "save_meta_data", False, "unused"),
    (_I, "gpu_device_id", 0, "HIP device the engine runs on"),
    # ---- frozen graphs (args.py:97-98): weights from the Const nodes of a frozen GraphDef (dcscn-super-resolution_amd/frozen.py)
    (_B, "frozenInference", False, "Flag for whether the model to evaluate is frozen."),
    (_S, "frozen_graph_path", "./model_to_freeze/frozen_model_optimized.pb", "the path to a frozen model if performing inference from it"),
]
for _define, _name, _default, _help in _TABLE:
    _define(_name, _default, _help)


def get():
    print("Python Interpreter version:%s" % sys.version[:3])
    print("dcscn_amd engine: %s" % dcscn_amd.engine.library_path())
    print("numpy version:%s" % np.__version__)
    return FLAGS
