"""``SuperResolution``: the reference's model object (DCSCN.py:28-769) with its inference surface
re-hosted on the MI355X engine.

What stays identical for a caller (sr.py:38-47, evaluate.py:44-107): constructor arguments, the
model-name derivation, ``build_graph / build_optimizer / build_summary_saver / init_all_variables /
load_model``, ``do / do_for_file / do_for_evaluate / do_for_evaluate_with_output /
evaluate_bicubic / evaluate``, their return values, printed lines and output file layout.

What changes underneath: ``build_graph`` derives the layer list instead of a TF graph,
``load_model`` reads the TF checkpoint without TensorFlow (ckpt.py) and uploads it to the HIP
engine, and ``do`` calls the C ABI (engine.py) where the reference calls ``sess.run``.  The
self-ensemble copies run as two device batches instead of eight sequential batch-1 runs.
Training (train.py, DCSCN.py:334-532) is out of scope.
"""

import logging
import math
import os
import sys

import numpy as np

from . import ckpt, engine, frozen, imaging as util

BICUBIC_METHOD_STRING = "bicubic"


def build_input_image(image, width=0, height=0, channels=1, scale=1, alignment=0, convert_ycbcr=True):
    """Crop / align / convert / downscale a loaded image into the network input (loader.py:42-67)."""
    if width != 0 and height != 0:
        if image.shape[0] != height or image.shape[1] != width:
            x = (image.shape[1] - width) // 2
            y = (image.shape[0] - height) // 2
            image = image[y: y + height, x: x + width, :]
    if alignment > 1:
        image = util.set_image_alignment(image, alignment)
    if channels == 1 and image.shape[2] == 3:
        if convert_ycbcr:
            image = util.convert_rgb_to_y(image)
    elif convert_ycbcr:
        image = util.convert_rgb_to_ycbcr(image)
    if scale != 1:
        image = util.resize_image_by_pil(image, 1.0 / scale)
    return image


class SuperResolution:
    def __init__(self, flags, model_name=""):
        # graph settings (helper/tf_graph.py:19-63)
        self.dropout_rate = flags.dropout_rate
        self.activator = flags.activator
        self.batch_norm = flags.batch_norm
        self.cnn_size = flags.cnn_size
        self.cnn_stride = 1
        self.initializer = flags.initializer
        self.weight_dev = flags.weight_dev
        self.enable_log = flags.enable_log
        self.checkpoint_dir = flags.checkpoint_dir
        self.tf_log_dir = flags.tf_log_dir
        self.gpu_device_id = flags.gpu_device_id

        # model parameters (DCSCN.py:33-48)
        self.scale = flags.scale
        self.layers = flags.layers
        self.filters = flags.filters
        self.min_filters = min(flags.filters, flags.min_filters)
        self.filters_decay_gamma = flags.filters_decay_gamma
        self.use_nin = flags.use_nin
        self.nin_filters = flags.nin_filters
        self.nin_filters2 = flags.nin_filters2
        self.reconstruct_layers = max(flags.reconstruct_layers, 1)
        self.reconstruct_filters = flags.reconstruct_filters
        self.resampling_method = BICUBIC_METHOD_STRING
        self.pixel_shuffler = flags.pixel_shuffler
        self.pixel_shuffler_filters = flags.pixel_shuffler_filters
        self.self_ensemble = flags.self_ensemble
        self.ensemble_group = None          # evaluate.py: a shard.Group when the ensemble transforms of an image are spread over ranks
        self.depthwise_separable = flags.depthwise_separable

        # image processing parameters (DCSCN.py:76-82)
        self.max_value = flags.max_value
        self.channels = flags.channels
        self.output_channels = 1
        self.psnr_calc_border_size = flags.psnr_calc_border_size
        if self.psnr_calc_border_size < 0:
            self.psnr_calc_border_size = self.scale

        self.name = self.get_model_name(model_name)

        # bookkeeping the reference logs from build_graph
        self.features = ""
        self.receptive_fields = 0
        self.complexity = 0
        self.legacy_no_c = False
        self._engine = None
        self._weights = None
        self._pending_init = None

        util.make_dir(self.checkpoint_dir)
        util.set_logging(flags.log_filename, stream_log_level=logging.INFO, file_log_level=logging.INFO)
        logging.info("\nDCSCN v2-------------------------------------")
        logging.info("%s [%s]" % (util.get_now_date(), self.name))

    # ---- naming (DCSCN.py:108-144) -----------------------------------------------------------
    def get_model_name(self, model_name, name_postfix=""):
        if model_name != "":
            return "dcscn_%s" % model_name
        name = "dcscn_L%d_F%d" % (self.layers, self.filters)
        if self.min_filters != 0:
            name += "to%d" % self.min_filters
        if self.filters_decay_gamma != 1.5:
            name += "_G%2.2f" % self.filters_decay_gamma
        if self.cnn_size != 3:
            name += "_C%d" % self.cnn_size
        if self.scale != 2:
            name += "_Sc%d" % self.scale
        if self.use_nin:
            name += "_NIN"
            if self.nin_filters != 0:
                name += "_A%d" % self.nin_filters
            if self.nin_filters2 != self.nin_filters // 2:
                name += "_B%d" % self.nin_filters2
        if self.pixel_shuffler:
            name += "_PS"
        if self.max_value != 255.0:
            name += "_M%2.1f" % self.max_value
        if self.activator != "prelu":
            name += "_%s" % self.activator
        if self.batch_norm:
            name += "_BN"
        if self.depthwise_separable:
            name += "_DS"
        if self.reconstruct_layers >= 1:
            name += "_R%d" % self.reconstruct_layers
            if self.reconstruct_filters != 1:
                name += "F%d" % self.reconstruct_filters
        if name_postfix != "":
            name += "_" + name_postfix
        return name

    # ---- graph -------------------------------------------------------------------------------
    def _engine_config(self):
        return dict(scale=self.scale, layers=self.layers, filters=self.filters, min_filters=self.min_filters,
                    filters_decay_gamma=self.filters_decay_gamma, cnn_size=self.cnn_size, use_nin=self.use_nin,
                    nin_filters=self.nin_filters, nin_filters2=self.nin_filters2,
                    reconstruct_layers=self.reconstruct_layers, reconstruct_filters=self.reconstruct_filters,
                    activator=self.activator, pixel_shuffler=self.pixel_shuffler,
                    pixel_shuffler_filters=self.pixel_shuffler_filters,
                    depthwise_separable=self.depthwise_separable, channels=self.channels,
                    legacy_no_c=self.legacy_no_c, batch_norm=self.batch_norm)

    def _create_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine = engine.Engine(self._engine_config(), device=self.gpu_device_id)
        self._weights = None
        return self._engine

    def build_graph(self):
        """Instantiate the layer plan on the device and log what the reference logs (DCSCN.py:331-332)."""
        eng = self._create_engine()
        self._log_graph(eng)

    def _log_graph(self, eng):
        layers = eng.layers()
        feats, total, rf, complexity = [], 0, 0, 0
        for i, li in enumerate(layers):
            cout, k, cin = li["out_channels"], li["kernel_size"], li["in_channels"]
            feats.append(cout)
            if i < self.layers:
                total += cout
            # "Complexity" counts every layer at LR resolution (tf_graph.py:100,106,110)
            complexity += (k * k * cin + cin * cout) if li["depthwise_separable"] else k * k * cin * cout
            complexity += cout if li["has_bias"] else 0
            complexity += cout if li["activator"] != 0 else 0
            rf = k if rf == 0 else rf + (k - 1)
        if self.use_nin:
            rf -= (self.cnn_size - 1)                         # DCSCN.py:275
        self.features = " ".join("%d" % f for f in feats[:self.layers]) + " Total: (%d) " % total \
                        + " ".join("%d" % f for f in feats[self.layers:]) + " "
        self.complexity = complexity
        self.receptive_fields = rf
        logging.info("Feature:%s Complexity:%s Receptive Fields:%d" % (
            self.features, "{:,}".format(self.complexity), self.receptive_fields))

    def build_optimizer(self):
        """Training-only in the reference (DCSCN.py:334-395); inference needs nothing here."""

    def build_summary_saver(self, with_saver=True):
        """TensorBoard writers / Saver in the reference (tf_graph.py:298-305); nothing to create."""

    def init_all_variables(self):
        """``tf.global_variables_initializer`` (tf_graph.py:73-75): He truncated-normal filters
        (utilty.py:360-363), zero bias, PReLU slope 0.1 -- replaced by ``load_model`` for real use."""
        if self._engine is None:
            self.build_graph()
        rng = np.random.default_rng()
        tensors = {}
        for name, shape in self._engine.tensor_specs():
            leaf = name.rsplit("/", 1)[-1]
            if leaf in ("conv_W", "depthwise_W", "pointwise_W"):
                std = math.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))
                tensors[name] = np.clip(rng.standard_normal(shape), -2.0, 2.0).astype(np.float32) * np.float32(std)
            elif leaf == "conv_B":
                tensors[name] = np.zeros(shape, np.float32)
            else:
                tensors[name] = np.full(shape, 0.1, np.float32)
        self._pending_init = tensors
        print("Model initialized.")

    # ---- checkpoint (tf_graph.py:263-280) ------------------------------------------------------
    def load_model(self, name="", trial=0, output_log=False):
        if name == "" or name == "default":
            name = self.name
        if trial > 0:
            filename = self.checkpoint_dir + "/" + name + "_" + str(trial) + ".ckpt"
        else:
            filename = self.checkpoint_dir + "/" + name + ".ckpt"
        if not os.path.isfile(filename + ".index"):
            print("Error. [%s] is not exist!" % filename)
            sys.exit(-1)
        if not ckpt.has_data(filename):
            print("Error. [%s] has no data shard (.data-00000-of-00001)!" % filename)
            sys.exit(-1)
        tensors = ckpt.load_checkpoint(filename)
        self.load_weights(tensors)
        if output_log:
            logging.info("Model restored [ %s ]." % filename)
        else:
            print("Model restored [ %s ]." % filename)

    def save_model(self, name="", trial=0, output_log=False):
        """tf.train.Saver.save (tf_graph.py:282-296): the model's variables as a TF V2 checkpoint
        (``<checkpoint_dir>/<name>[_<trial>].ckpt.{index,data-00000-of-00001}`` + the ``checkpoint`` state file), written
        without TensorFlow (ckpt.save_checkpoint; byte-identical to what the reference's Saver writes for the same tensors)."""
        if name == "" or name == "default":
            name = self.name
        if trial > 0:
            filename = self.checkpoint_dir + "/" + name + "_" + str(trial) + ".ckpt"
        else:
            filename = self.checkpoint_dir + "/" + name + ".ckpt"
        tensors = self._weights if self._weights is not None else self._pending_init
        if tensors is None:
            self.init_all_variables()
            tensors = self._pending_init
        ckpt.save_checkpoint(filename, tensors)
        if output_log:
            logging.info("Model saved [%s]." % filename)
        else:
            print("Model saved [%s]." % filename)

    def load_graph(self, frozen_graph_filename="./model_to_freeze/frozen_model_optimized.pb"):
        """--frozenInference (DCSCN.py:192-220): the weights come from the Const nodes of a frozen GraphDef (variables keep
        their checkpoint names through freeze_graph / optimize_for_inference); the topology is the one the flags describe,
        and a file that does not fit it is rejected by the engine (missing variable / shape mismatch)."""
        if not os.path.isfile(frozen_graph_filename):
            print("Error. [%s] is not exist!" % frozen_graph_filename)
            sys.exit(-1)
        tensors = frozen.read_frozen_graph(frozen_graph_filename)
        if not any(k.endswith(("/conv_W", "/pointwise_W")) for k in tensors):
            tensors = frozen.read_frozen_graph(frozen_graph_filename, prefix="prefix/")     # a re-exported import_graph_def copy
        self.load_weights({k: v for k, v in tensors.items() if not ckpt.is_optimizer_slot(k)})
        print("Frozen graph loaded [ %s ]." % frozen_graph_filename)

    def load_weights(self, tensors):
        """Upload ``{variable name: ndarray}``.  Checkpoints written by the reference's older graph
        (the shipped dcscn_L2_* files: use_nin=False and no 1x1 "C" layer) are recognised by their
        variable set and get the matching topology."""
        legacy = (not self.use_nin) and ("C/conv_W" not in tensors) and ("C/pointwise_W" not in tensors)
        if legacy != self.legacy_no_c or self._engine is None or self._engine.finalized:
            self.legacy_no_c = legacy
            self._create_engine()
        self._engine.load_weights(tensors)
        self._weights = tensors
        self._pending_init = None

    def _ready_engine(self):
        if self._engine is None:
            self.build_graph()
        if not self._engine.finalized:
            pending = self._pending_init
            if pending is None:
                self.init_all_variables()
                pending = self._pending_init
            self._engine.load_weights(pending)
            self._pending_init = None
        return self._engine

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # ---- inference (DCSCN.py:547-586) ----------------------------------------------------------
    def do(self, input_image, bicubic_input_image=None):
        h, w = input_image.shape[:2]
        ch = input_image.shape[2] if len(input_image.shape) > 2 else 1
        if ch != 1:
            raise ValueError("do() expects a single-channel image, got %d channels" % ch)
        # the device bicubic reproduces Pillow's mode-'F' path (float images).  A uint8 single-channel image goes through
        # Pillow's mode 'L' in the reference (utilty.py:229-232: rounded and clipped to 0..255), so it stays on the host
        if bicubic_input_image is None and self.resampling_method == "bicubic" and np.issubdtype(np.asarray(input_image).dtype, np.floating):
            # DCSCN.py:552-554 on the device: dcscn_resize_bicubic is bit-compatible with Pillow's mode-'F' BICUBIC
            # (tests/test_resize_hip.py), so x2 never has to be built or uploaded by the host
            eng = self._ready_engine()
            x = np.ascontiguousarray(input_image, dtype=np.float32).reshape(h, w)
            if self.max_value == 255.0 and self.self_ensemble <= 1:
                return eng.forward_lr(x[None])[0]
            bicubic_input_image = eng.resize_bicubic(x, self.scale * h, self.scale * w).reshape(self.scale * h, self.scale * w, 1)
        elif bicubic_input_image is None:
            bicubic_input_image = util.resize_image_by_pil(input_image, self.scale,
                                                           resampling_method=self.resampling_method)
        if self.max_value != 255.0:
            input_image = np.multiply(input_image, self.max_value / 255.0)
            bicubic_input_image = np.multiply(bicubic_input_image, self.max_value / 255.0)

        eng = self._ready_engine()
        x = np.ascontiguousarray(input_image, dtype=np.float32).reshape(h, w)
        x2 = np.ascontiguousarray(bicubic_input_image, dtype=np.float32).reshape(self.scale * h, self.scale * w)
        if self.self_ensemble > 1 and self.ensemble_group is not None:
            # (image, transform) work items sharded over the ranks (evaluate.py); float64 mean in the reference's order
            output = self.ensemble_group.ensemble_mean(
                x[:, :, None], x2[:, :, None], self.self_ensemble,
                lambda a, b: eng.forward(a[None], b[None])[0], util.flip)
            if output is None:
                return None                                   # not rank 0: the mean (and everything computed from it) lives there
        elif self.self_ensemble > 1:
            output = eng.forward_ensemble(x, x2, self.self_ensemble)            # float64, like np.zeros + +=
        else:
            output = eng.forward(x[None, :, :, None], x2[None, :, :, None])[0]   # float32, like sess.run
        if self.max_value != 255.0:
            return np.multiply(output, 255.0 / self.max_value)
        return output

    def do_batch(self, lr_batch, bicubic_batch):
        """Extension: [n, h, w, 1] + [n, s*h, s*w, 1] float32 -> [n, s*h, s*w, 1] in one device pass."""
        return self._ready_engine().forward(lr_batch, bicubic_batch)

    @util.with_deferred_saves                                # the PNGs are encoded on worker threads; on disk when this returns
    def do_for_file(self, file_path, output_folder="output"):
        """sr.py's work (DCSCN.py:588-614): writes the original, bicubic and SR images."""
        org_image = util.load_image(file_path)
        filename, extension = os.path.splitext(os.path.basename(file_path))
        output_folder += "/" + self.name + "/"
        util.save_image(output_folder + filename + extension, org_image)

        scaled_image = util.resize_image_by_pil(org_image, self.scale, resampling_method=self.resampling_method)
        util.save_image(output_folder + filename + "_bicubic" + extension, scaled_image)

        if len(org_image.shape) >= 3 and org_image.shape[2] == 3 and self.channels == 1:
            input_y_image = util.convert_rgb_to_y(org_image)
            scaled_image = util.resize_image_by_pil(input_y_image, self.scale, resampling_method=self.resampling_method)
            util.save_image(output_folder + filename + "_bicubic_y" + extension, scaled_image)
            if self._device_colour_path(org_image):
                # Y, x2, the forward pass and the YCbCr -> RGB recombination in one device pipeline (dcscn_sr_rgb)
                scaled_rgb = util.resize_image_by_pil(org_image, self.scale, self.resampling_method)
                output_y_image, image = self._ready_engine().sr_rgb(org_image, scaled_rgb, self.self_ensemble)
                util.save_image(output_folder + filename + "_result_y" + extension, output_y_image)
            else:
                output_y_image = self.do(input_y_image)
                util.save_image(output_folder + filename + "_result_y" + extension, output_y_image)
                scaled_ycbcr_image = util.convert_rgb_to_ycbcr(
                    util.resize_image_by_pil(org_image, self.scale, self.resampling_method))
                image = util.convert_y_and_cbcr_to_rgb(output_y_image, scaled_ycbcr_image[:, :, 1:3])
        else:
            scaled_image = util.resize_image_by_pil(org_image, self.scale, resampling_method=self.resampling_method)
            util.save_image(output_folder + filename + "_bicubic_y" + extension, scaled_image)
            image = self.do(org_image)
        util.save_image(output_folder + filename + "_result" + extension, image)

    def _device_colour_path(self, image):
        """uint8 RGB through the device colour / bicubic kernels: they reproduce the float64 numpy colour math and
        Pillow's mode-'F' BICUBIC only (other resampling methods, other value ranges stay on the host path)."""
        return (self.resampling_method == BICUBIC_METHOD_STRING and self.max_value == 255.0 and self.channels == 1
                and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3
                and self.ensemble_group is None)              # the all-device pipeline runs the whole ensemble on this rank

    def _evaluation_inputs(self, file_path):
        """(true image aligned, true Y or grey, LR input, bicubic of LR) -- DCSCN.py:674-683."""
        true_image = util.set_image_alignment(util.load_image(file_path, print_console=False), self.scale)
        if true_image.shape[2] == 3 and self.channels == 1:
            input_image = build_input_image(true_image, channels=self.channels, scale=self.scale,
                                            alignment=self.scale, convert_ycbcr=True)
            true_y = util.convert_rgb_to_y(true_image)
        elif true_image.shape[2] == 1 and self.channels == 1:
            input_image = build_input_image(true_image, channels=self.channels, scale=self.scale, alignment=self.scale)
            true_y = true_image
        else:
            return true_image, None, None, None
        bicubic = util.resize_image_by_pil(input_image, self.scale, resampling_method=self.resampling_method)
        return true_image, true_y, input_image, bicubic

    def do_for_evaluate(self, file_path, print_console=False):
        """(psnr, ssim) of one file (DCSCN.py:672-703)."""
        true_image = util.set_image_alignment(util.load_image(file_path, print_console=False), self.scale)
        if self._device_colour_path(true_image):
            # one upload: Y conversion, both bicubic resizes, the (ensemble of) forward pass on the device (dcscn_evaluate_rgb)
            true_y, output = self._ready_engine().evaluate_rgb(true_image, self.self_ensemble)
        else:
            true_image, true_y, input_image, bicubic = self._evaluation_inputs(file_path)
            if true_y is None:
                return None, None
            output = self.do(input_image, bicubic)
            if output is None:
                return 0.0, 0.0                               # split ensemble, not rank 0: the values are rank 0's (evaluate.py logs there)
        psnr, ssim = util.compute_psnr_and_ssim(true_y, output, border_size=self.psnr_calc_border_size)
        if print_console:
            print("[%s] PSNR:%f, SSIM:%f" % (file_path, psnr, ssim))
        return psnr, ssim

    def do_for_evaluate_many(self, file_paths, print_console=False):
        """[(psnr, ssim, seconds)] of do_for_evaluate over the files, in order, as a three-stage pipeline: the
        reference's loop (evaluate.py:89-107) is decode -> network -> PSNR / SSIM, strictly serial, and on this path the device is
        ~5 % of an image's time (profiles/r03_c4_eval_profile.txt).  Here the PNG decode of the NEXT files runs on one worker thread,
        the metric code (utilty.py:509-536, numpy / scipy: it releases the GIL) of the PREVIOUS ones on two more, and this thread
        alone drives the engine (the handle is not thread safe) in file order.  Every value is computed by the same functions on
        the same data as in do_for_evaluate: the results are identical.  `seconds` = wall clock of the whole call / number of files
        (what a caller waits per file; the reference's figure is the serial time of each file)."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        n = len(file_paths)
        if n == 0:
            return []
        t0 = time.time()

        def decode(path):
            return util.set_image_alignment(util.load_image(path, print_console=False), self.scale)

        def metrics(true_y, output):
            return util.compute_psnr_and_ssim(true_y, output, border_size=self.psnr_calc_border_size)

        with ThreadPoolExecutor(max_workers=1) as dec, ThreadPoolExecutor(max_workers=2) as met:
            ahead = 2
            pending = {i: dec.submit(decode, file_paths[i]) for i in range(min(ahead, n))}
            results = []
            for i in range(n):
                true_image = pending.pop(i).result()
                if i + ahead < n:
                    pending[i + ahead] = dec.submit(decode, file_paths[i + ahead])
                if self._device_colour_path(true_image):
                    true_y, output = self._ready_engine().evaluate_rgb(true_image, self.self_ensemble)
                else:
                    true_image, true_y, input_image, bicubic = self._evaluation_inputs(file_paths[i])
                    output = None if true_y is None else self.do(input_image, bicubic)
                results.append(None if true_y is None else met.submit(metrics, true_y, output))
            values = [(None, None) if r is None else r.result() for r in results]
        per_file = (time.time() - t0) / n
        if print_console:
            for path, (psnr, ssim) in zip(file_paths, values):
                print("[%s] PSNR:%s, SSIM:%s" % (path, psnr, ssim))
        return [(psnr, ssim, per_file) for psnr, ssim in values]

    @util.with_deferred_saves
    def do_for_evaluate_with_output(self, file_path, output_directory, print_console=False):
        """As above, also writing the result images (DCSCN.py:616-670)."""
        filename, extension = os.path.splitext(file_path)
        output_directory += "/" + self.name + "/"
        util.make_dir(output_directory)

        true_image = util.set_image_alignment(util.load_image(file_path, print_console=False), self.scale)
        input_image = util.resize_image_by_pil(true_image, 1.0 / self.scale, resampling_method=self.resampling_method)
        input_bicubic_image = util.resize_image_by_pil(input_image, self.scale, resampling_method=self.resampling_method)
        util.save_image(output_directory + filename + "_input_bicubic" + extension, input_bicubic_image)

        if true_image.shape[2] == 3 and self.channels == 1:
            input_y_image = build_input_image(true_image, channels=self.channels, scale=self.scale,
                                              alignment=self.scale, convert_ycbcr=True)
            input_bicubic_y_image = util.resize_image_by_pil(input_y_image, self.scale,
                                                             resampling_method=self.resampling_method)
            true_ycbcr_image = util.convert_rgb_to_ycbcr(true_image)
            output_y_image = self.do(input_y_image, input_bicubic_y_image)
            psnr, ssim = util.compute_psnr_and_ssim(true_ycbcr_image[:, :, 0:1], output_y_image,
                                                    border_size=self.psnr_calc_border_size)
            loss_image = util.get_loss_image(true_ycbcr_image[:, :, 0:1], output_y_image,
                                             border_size=self.psnr_calc_border_size)
            output_color_image = util.convert_y_and_cbcr_to_rgb(output_y_image, true_ycbcr_image[:, :, 1:3])
            util.save_image(output_directory + file_path, true_image)
            util.save_image(output_directory + filename + "_input" + extension, input_y_image)
            util.save_image(output_directory + filename + "_input_bicubic_y" + extension, input_bicubic_y_image)
            util.save_image(output_directory + filename + "_true_y" + extension, true_ycbcr_image[:, :, 0:1])
            util.save_image(output_directory + filename + "_result" + extension, output_y_image)
            util.save_image(output_directory + filename + "_result_c" + extension, output_color_image)
            util.save_image(output_directory + filename + "_loss" + extension, loss_image)
        elif true_image.shape[2] == 1 and self.channels == 1:
            input_image = build_input_image(true_image, channels=self.channels, scale=self.scale, alignment=self.scale)
            input_bicubic_y_image = util.resize_image_by_pil(input_image, self.scale,
                                                             resampling_method=self.resampling_method)
            output_image = self.do(input_image, input_bicubic_y_image)
            psnr, ssim = util.compute_psnr_and_ssim(true_image, output_image, border_size=self.psnr_calc_border_size)
            util.save_image(output_directory + file_path, true_image)
            util.save_image(output_directory + filename + "_result" + extension, output_image)
        else:
            return None, None

        if print_console:
            print("[%s] PSNR:%f, SSIM:%f" % (filename, psnr, ssim))
        return psnr, ssim

    def evaluate_bicubic(self, file_path, print_console=False):
        """PSNR / SSIM of plain bicubic upscaling (DCSCN.py:705-725)."""
        true_image, true_y, input_image, bicubic = self._evaluation_inputs(file_path)
        if true_y is None:
            return None, None
        psnr, ssim = util.compute_psnr_and_ssim(true_y, bicubic, border_size=self.psnr_calc_border_size)
        if print_console:
            print("PSNR:%f, SSIM:%f" % (psnr, ssim))
        return psnr, ssim

    def evaluate(self, test_filenames):
        """Mean (psnr, ssim) over files (DCSCN.py:534-545)."""
        if len(test_filenames) == 0:
            return 0, 0
        total_psnr = total_ssim = 0
        for filename in test_filenames:
            psnr, ssim = self.do_for_evaluate(filename, print_console=False)
            total_psnr += psnr
            total_ssim += ssim
        return total_psnr / len(test_filenames), total_ssim / len(test_filenames)
