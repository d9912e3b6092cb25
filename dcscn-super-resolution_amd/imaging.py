"""Host-side image helpers of the evaluation / SR recipe (what helper/utilty.py provides to
DCSCN.py, sr.py and evaluate.py), without TensorFlow, imageio, scikit-image or scipy.misc.

Only the functions the inference path calls are provided; citations are into the reference tree.
Everything here is glue that stays on the host: file I/O through Pillow, colour conversion and
metrics in numpy/scipy.  Bicubic resampling deliberately goes through Pillow exactly as the
reference does (utilty.py:211-239) because x2 = bicubic(LR) feeds the network's residual add and
must match bit for bit.
"""

import datetime
import logging
import math
import os

import numpy as np
from PIL import Image


class LoadError(Exception):
    """utilty.py:51-53."""

    def __init__(self, message):
        super().__init__(message)
        self.message = message


# ---- filesystem ---------------------------------------------------------------------------------

def make_dir(directory):
    if not os.path.exists(directory):
        os.makedirs(directory)


def get_files_in_directory(path):
    """Files (not directories, not dot-files) in ``os.listdir`` order, utilty.py:67-71."""
    if not path.endswith("/"):
        path = path + "/"
    return [path + f for f in os.listdir(path) if os.path.isfile(os.path.join(path, f)) and not f.startswith(".")]


def clean_dir(path):
    if not os.path.isdir(path):
        return
    for entry in os.listdir(path):
        full = os.path.join(path, entry)
        try:
            if os.path.isfile(full):
                os.remove(full)
            elif os.path.isdir(full):
                clean_dir(full)
                os.rmdir(full)
        except OSError as error:
            print("OS error: {0}".format(error))


def get_now_date():
    d = datetime.datetime.today()
    return "%s/%s/%s %s:%s:%s" % (d.year, d.month, d.day, d.hour, d.minute, d.second)


def set_logging(filename, stream_log_level=logging.INFO, file_log_level=logging.INFO, tf_log_level=None):
    """Root logger to the console and ``filename`` (utilty.py:97-110); the TF verbosity is ignored."""
    logger = logging.getLogger()
    logger.handlers = []
    stream = logging.StreamHandler()
    stream.setLevel(stream_log_level)
    logger.addHandler(stream)
    if filename:
        file_log = logging.FileHandler(filename=filename)
        file_log.setLevel(file_log_level)
        logger.addHandler(file_log)
    logger.setLevel(min(stream_log_level, file_log_level))


# ---- image files --------------------------------------------------------------------------------

def load_image(filename, width=0, height=0, channels=0, alignment=0, print_console=True):
    """uint8 [H, W, C] array; alpha planes dropped (utilty.py:242-266, imageio replaced by Pillow)."""
    if not os.path.isfile(filename):
        raise LoadError("File not found [%s]" % filename)
    try:
        with Image.open(filename) as im:
            if im.mode in ("P", "CMYK", "YCbCr", "1"):
                im = im.convert("RGBA" if "transparency" in im.info else "RGB")
            elif im.mode in ("I;16", "I", "F"):
                im = im.convert("L")
            image = np.atleast_3d(np.array(im))
    except (OSError, ValueError) as exc:
        raise LoadError("Cannot read image [%s]: %s" % (filename, exc))

    if (width != 0 and image.shape[1] != width) or (height != 0 and image.shape[0] != height):
        raise LoadError("Attributes mismatch")
    if channels != 0 and image.shape[2] != channels:
        raise LoadError("Attributes mismatch")
    if alignment != 0 and ((width % alignment) != 0 or (height % alignment) != 0):
        raise LoadError("Attributes mismatch")
    if image.shape[2] == 2:          # grey + alpha
        image = image[:, :, 0:1]
    elif image.shape[2] >= 4:
        image = image[:, :, 0:3]
    if print_console:
        print("Loaded [%s]: %d x %d x %d" % (filename, image.shape[1], image.shape[0], image.shape[2]))
    return image


# PNG encoding is the most expensive thing evaluate.py / sr.py do per image with --save_results (7 files, ~225 ms of zlib per Set14
# image against ~10 ms for everything else): the files are encoded and written on worker threads (Pillow releases the GIL).  The pixel
# data is fixed at the call (the uint8 cast is done here, like the reference does it); a caller's method returns only after ITS files
# are on disk unless it runs inside `deferred_saves()` (evaluate.py: the whole data set is one batch of writes).
_save_pool = None
_save_pending = []
_save_failed = []             # writes that failed while save_image was only waiting for room: raised by flush_saves
_save_defer = 0
_SAVE_MAX_PENDING = 64        # images in flight: a fast device path must not pile up uint8 copies faster than they are encoded


def _encode_and_write(filename, image, mode):
    Image.fromarray(image, mode=mode).save(filename)


def save_image(filename, image, print_console=True):
    """Cast to uint8 the way the reference does (plain ``astype``, no rounding: utilty.py:113-130).

    Synchronous, like the reference's -- except inside a ``deferred_saves()`` block, where the PNG is encoded and written on a
    worker thread and the block's exit waits for it (model.do_for_file / do_for_evaluate_with_output write up to seven images per file).
    """
    global _save_pool
    if len(image.shape) >= 3 and image.shape[2] == 1:
        image = image.reshape(image.shape[0], image.shape[1])
    directory = os.path.dirname(filename)
    if directory != "" and not os.path.exists(directory):
        os.makedirs(directory, exist_ok=True)
    with np.errstate(invalid="ignore"):
        image = np.ascontiguousarray(image.astype(np.uint8))
    mode = "RGB" if image.ndim == 3 and image.shape[2] == 3 else None
    if _save_defer > 0:
        if _save_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _save_pool = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1))
        while len(_save_pending) >= _SAVE_MAX_PENDING:      # back-pressure: wait for the oldest write
            oldest = _save_pending.pop(0)
            if oldest.exception() is not None:              # (waits) a failed write is reported by flush_saves at the block's
                _save_failed.append(oldest)                 # exit, not from inside this unrelated call
        _save_pending.append(_save_pool.submit(_encode_and_write, filename, image, mode))
    else:
        _encode_and_write(filename, image, mode)
    if print_console:
        print(("Queued [%s]" if _save_defer > 0 else "Saved [%s]") % filename)


def flush_saves():
    """Wait until every image handed to save_image inside a deferred_saves() block is on disk (re-raises the first write error)."""
    pending, _save_pending[:] = _save_failed + list(_save_pending), []
    _save_failed[:] = []
    first = None
    for f in pending:                                       # wait for ALL of them, then raise the first error
        err = f.exception()
        if err is not None and first is None:
            first = err
    if first is not None:
        raise first


class deferred_saves(object):
    """``with deferred_saves():`` -- save_image calls inside return at once, the (outermost) block's exit waits for all of them."""
    def __enter__(self):
        global _save_defer
        _save_defer += 1
        return self

    def __exit__(self, exc_type, exc, tb):
        global _save_defer
        _save_defer -= 1
        if _save_defer == 0:
            try:
                flush_saves()
            except Exception:
                if exc_type is None:                          # (an exception from the block itself wins over a write error)
                    raise
                # ... but the write error is not lost: it goes to the log
                logging.exception("an image write failed while the block was already unwinding from %s", exc_type.__name__)
        return False


def with_deferred_saves(fn):
    """Decorator: the function's save_image calls run on worker threads; it returns when its files are on disk."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with deferred_saves():
            return fn(*args, **kwargs)
    return wrapper


# ---- colour -------------------------------------------------------------------------------------

_Y_ROW = np.array([[65.738 / 256.0, 129.057 / 256.0, 25.064 / 256.0]])
_YCBCR = np.array([[65.738 / 256.0, 129.057 / 256.0, 25.064 / 256.0],
                   [-37.945 / 256.0, -74.494 / 256.0, 112.439 / 256.0],
                   [112.439 / 256.0, -94.154 / 256.0, -18.285 / 256.0]])
_RGB = np.array([[298.082 / 256.0, 0, 408.583 / 256.0],
                 [298.082 / 256.0, -100.291 / 256.0, -208.120 / 256.0],
                 [298.082 / 256.0, 516.412 / 256.0, 0]])


def convert_rgb_to_y(image):
    """ITU-R BT.601 luma in float64, not rounded (utilty.py:142-149)."""
    if len(image.shape) <= 2 or image.shape[2] == 1:
        return image
    return image.dot(_Y_ROW.T) + 16.0


def convert_rgb_to_ycbcr(image):
    """utilty.py:152-165."""
    if len(image.shape) < 2 or image.shape[2] == 1:
        return image
    out = image.dot(_YCBCR.T)
    out[:, :, 0] += 16.0
    out[:, :, [1, 2]] += 128.0
    return out


def convert_ycbcr_to_rgb(ycbcr_image):
    """utilty.py:168-179."""
    shifted = np.zeros([ycbcr_image.shape[0], ycbcr_image.shape[1], 3])
    shifted[:, :, 0] = ycbcr_image[:, :, 0] - 16.0
    shifted[:, :, [1, 2]] = ycbcr_image[:, :, [1, 2]] - 128.0
    return shifted.dot(_RGB.T)


def convert_y_and_cbcr_to_rgb(y_image, cbcr_image):
    """utilty.py:182-193."""
    if len(y_image.shape) <= 2:
        y_image = y_image.reshape(y_image.shape[0], y_image.shape[1], 1)
    if len(y_image.shape) == 3 and y_image.shape[2] == 3:
        y_image = y_image[:, :, 0:1]
    ycbcr = np.zeros([y_image.shape[0], y_image.shape[1], 3])
    ycbcr[:, :, 0] = y_image[:, :, 0]
    ycbcr[:, :, 1:3] = cbcr_image[:, :, 0:2]
    return convert_ycbcr_to_rgb(ycbcr)


# ---- geometry -----------------------------------------------------------------------------------

def set_image_alignment(image, alignment):
    """Crop bottom/right to a multiple of ``alignment``; drop alpha (utilty.py:196-208)."""
    alignment = int(alignment)
    height = (image.shape[0] // alignment) * alignment
    width = (image.shape[1] // alignment) * alignment
    if image.shape[1] != width or image.shape[0] != height:
        image = image[:height, :width, :]
    if len(image.shape) >= 3 and image.shape[2] >= 4:
        image = image[:, :, 0:3]
    return image


_PIL_METHODS = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}


def resize_image_by_pil(image, scale, resampling_method="bicubic"):
    """Pillow resize (utilty.py:211-239): RGB stays uint8 'RGB'; single-channel arrays keep their
    dtype's Pillow mode (float64 -> mode 'F' float32, uint8 -> 'L')."""
    height, width = image.shape[0], image.shape[1]
    new_width = int(width * scale)
    new_height = int(height * scale)
    method = _PIL_METHODS.get(resampling_method, Image.LANCZOS)
    if len(image.shape) == 3 and image.shape[2] in (3, 4):
        rgb = np.ascontiguousarray(image[:, :, 0:3]) if image.shape[2] == 4 else image
        im = Image.fromarray(rgb.astype(np.uint8) if rgb.dtype != np.uint8 else rgb, "RGB")
        return np.asarray(im.resize([new_width, new_height], resample=method))
    im = Image.fromarray(image.reshape(height, width))
    im = im.resize([new_width, new_height], resample=method)
    return np.asarray(im).reshape(new_height, new_width, 1)


def flip(image, flip_type, invert=False):
    """The eight self-ensemble transforms and their inverses (utilty.py:595-617)."""
    if flip_type == 0:
        return image
    if flip_type == 1:
        return np.flipud(image)
    if flip_type == 2:
        return np.fliplr(image)
    if flip_type == 3:
        return np.flipud(np.fliplr(image))
    if flip_type == 4:
        return np.rot90(image, -1 if invert else 1)
    if flip_type == 5:
        return np.rot90(image, 1 if invert else -1)
    if flip_type == 6:
        return np.rot90(np.flipud(image), -1) if invert else np.flipud(np.rot90(image))
    if flip_type == 7:
        return np.rot90(np.flipud(image), 1) if invert else np.flipud(np.rot90(image, -1))
    return None


# ---- metrics ------------------------------------------------------------------------------------

def trim_image_as_file(image):
    """Round, clip to [0, 255], float32 -- what a saved PNG would hold (utilty.py:501-506)."""
    image = np.clip(np.rint(image), 0, 255)
    return image if image.dtype == np.float32 else image.astype(np.float32)


def get_loss_image(image1, image2, scale=1.0, border_size=0):
    """Squared difference of the trimmed images, capped at 255 (utilty.py:480-498)."""
    if len(image1.shape) == 2:
        image1 = image1.reshape(image1.shape[0], image1.shape[1], 1)
    if len(image2.shape) == 2:
        image2 = image2.reshape(image2.shape[0], image2.shape[1], 1)
    if image1.shape != image2.shape:
        return None
    diff = np.subtract(trim_image_as_file(image1), trim_image_as_file(image2))
    loss = np.minimum(np.multiply(np.square(diff), scale), 255.0)
    if border_size > 0:
        loss = loss[border_size:-border_size, border_size:-border_size, :]
    return loss


def _psnr(a, b, data_range):
    """skimage.metrics.peak_signal_noise_ratio: float64 MSE, 10 log10(R^2 / mse)."""
    err = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, dtype=np.float64)
    if err == 0:
        return float("inf")
    return 10.0 * math.log10((data_range ** 2) / err)


def _ssim_gaussian(a, b, data_range, sigma, k1, k2):
    """skimage.metrics.structural_similarity(gaussian_weights=True, use_sample_covariance=True) for
    one N-D array pair: Gaussian window truncated at 3.5 sigma (11 taps for sigma 1.5), borders of
    (win-1)/2 excluded from the mean."""
    from scipy.ndimage import gaussian_filter
    truncate = 3.5
    radius = int(truncate * sigma + 0.5)
    win = 2 * radius + 1
    if any(s < win for s in a.shape):
        raise ValueError("win_size exceeds image extent")
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    npix = win ** a.ndim
    cov_norm = npix / (npix - 1.0)

    def f(x):
        return gaussian_filter(x, sigma=sigma, truncate=truncate)

    ux, uy = f(a), f(b)
    uxx, uyy, uxy = f(a * a), f(b * b), f(a * b)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    c1 = (k1 * data_range) ** 2
    c2 = (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    core = s[tuple(slice(pad, dim - pad) for dim in s.shape)]
    return float(core.mean(dtype=np.float64))


def compute_psnr_and_ssim(image1, image2, border_size=0):
    """PSNR and SSIM as the reference computes them (utilty.py:509-536).

    Quirk kept on purpose: the single-channel images are squeezed to 2-D and then handed to
    scikit-image with ``multichannel=True``, which makes it treat the LAST axis (image columns) as
    channels -- so the "SSIM" is the mean over columns of a 1-D SSIM along each column.
    """
    if len(image1.shape) == 2:
        image1 = image1.reshape(image1.shape[0], image1.shape[1], 1)
    if len(image2.shape) == 2:
        image2 = image2.reshape(image2.shape[0], image2.shape[1], 1)
    if image1.shape != image2.shape:
        return None
    image1 = trim_image_as_file(image1)
    image2 = trim_image_as_file(image2)
    if border_size > 0:
        image1 = image1[border_size:-border_size, border_size:-border_size, :]
        image2 = image2[border_size:-border_size, border_size:-border_size, :]
    if image1.shape[2] == 1:
        image1 = image1[:, :, 0]
        image2 = image2[:, :, 0]
    psnr = _psnr(image1, image2, 255)
    return psnr, _ssim_last_axis_channels(image1, image2, 255, 1.5, 0.01, 0.03)


def _ssim_last_axis_channels(a, b, data_range, sigma, k1, k2):
    """``structural_similarity(..., multichannel=True)``: the mean over the last axis of the SSIM of each
    ``a[..., c]``.  For the 2-D arrays the reference passes, every "channel" is one image column, i.e. a 1-D
    signal filtered along axis 0 -- done for all columns at once (same arithmetic as the per-column loop
    of ``_ssim_gaussian``; only the order of the final mean's additions differs)."""
    if a.ndim != 2:
        return float(np.mean([_ssim_gaussian(a[..., c], b[..., c], data_range, sigma, k1, k2)
                              for c in range(a.shape[-1])]))
    from scipy.ndimage import gaussian_filter1d
    truncate = 3.5
    radius = int(truncate * sigma + 0.5)
    win = 2 * radius + 1
    if a.shape[0] < win:
        raise ValueError("win_size exceeds image extent")
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    cov_norm = win / (win - 1.0)             # NP = win ** ndim with ndim = 1 per column

    def f(x):
        return gaussian_filter1d(x, sigma, axis=0, truncate=truncate)

    ux, uy = f(a), f(b)
    uxx, uyy, uxy = f(a * a), f(b * b), f(a * b)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    c1 = (k1 * data_range) ** 2
    c2 = (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    per_column = s[pad:a.shape[0] - pad, :].mean(axis=0, dtype=np.float64)
    return float(per_column.mean(dtype=np.float64))
