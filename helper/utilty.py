"""``helper.utilty`` of the reference, inference subset, without TensorFlow / imageio / scikit-image.
See dcscn-super-resolution_amd/imaging.py for the implementations and their reference citations."""

import dcscn_amd  # noqa: F401
from dcscn_amd.imaging import *          # noqa: F401,F403
from dcscn_amd.imaging import (LoadError, clean_dir, compute_psnr_and_ssim, convert_rgb_to_y,   # noqa: F401
                               convert_rgb_to_ycbcr, convert_y_and_cbcr_to_rgb, convert_ycbcr_to_rgb, flip,
                               get_files_in_directory, get_loss_image, get_now_date, load_image, make_dir,
                               resize_image_by_pil, save_image, flush_saves, deferred_saves, with_deferred_saves,
                               set_image_alignment, set_logging,
                               trim_image_as_file)
