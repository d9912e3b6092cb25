"""MI355X-native DCSCN super-resolution inference path.

Host side (Python) of the drop-in for the forward pass of jiny2001/dcscn-super-resolution:

  engine.py   ctypes binding of libdcscn_hip.so (the hand-written gfx950 kernels behind include/dcscn.h)
  ckpt.py     TensorFlow-free reader of the reference's V2 checkpoints
  build.py    hipcc build of the shared library
  model.py    SuperResolution: the reference's DCSCN.py inference surface on top of the engine
  imaging.py  image I/O, colour, Pillow bicubic, PSNR/SSIM (helper/utilty.py without TF/imageio/skimage)
  flags.py    absl/tf.app.flags-compatible flag registry (helper/args.py)
  shard.py    image/patch sharding over the GPUs of a node

The directory name is not a valid Python identifier; import it through the ``dcscn_amd`` alias module
at the repository root (``import dcscn_amd``) or ``importlib.import_module("dcscn-super-resolution_amd")``.
"""

__all__ = ["build", "ckpt", "engine", "flags", "imaging", "model", "shard"]
