"""``helper.loader`` of the reference, inference subset (loader.py:27-67).  The training datasets
(BatchDataSets / DynamicDataSets) are out of scope."""

import dcscn_amd  # noqa: F401
from dcscn_amd import imaging as util
from dcscn_amd.model import build_input_image          # noqa: F401


def build_image_set(file_path, channels=1, scale=1, convert_ycbcr=True, resampling_method="bicubic",
                    print_console=True):
    """(LR input, bicubic of LR, aligned true image), loader.py:27-38."""
    true_image = util.set_image_alignment(util.load_image(file_path, print_console=print_console), scale)
    if channels == 1 and true_image.shape[2] == 3 and convert_ycbcr:
        true_image = util.convert_rgb_to_y(true_image)
    input_image = util.resize_image_by_pil(true_image, 1.0 / scale, resampling_method=resampling_method)
    interpolated = util.resize_image_by_pil(input_image, scale, resampling_method=resampling_method)
    return input_image, interpolated, true_image


def load_input_image(filename, width=0, height=0, channels=1, scale=1, alignment=0, convert_ycbcr=True,
                     print_console=True):
    image = util.load_image(filename, print_console=print_console)
    return build_input_image(image, width, height, channels, scale, alignment, convert_ycbcr)
