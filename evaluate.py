"""Evaluate a trained model on a dataset -- same command line as the reference's evaluate.py:

    python evaluate.py --test_dataset=set5 [--scale=3] [--save_results=false] ...

Builds LR inputs from data/<test_dataset>/, super-resolves them on the MI355X and logs
``Model Average [<set>] PSNR:..., SSIM:..., Time (s): ...`` (evaluate.py:106-107).

One process drives one GPU (``--gpu_device_id``).  Launched under ``torch.distributed.run`` with N
ranks the work is partitioned as SURVEY.md 8(e) describes (the reference loops over files and, per file,
over the self-ensemble transforms: evaluate.py:89-107, DCSCN.py:559-573):

* at least 2 N images: whole images, assigned longest first by pixel count to the least-loaded rank;
* fewer (Set5 on 8 GPUs) and ``--self_ensemble`` > 1: (image, transform) work items -- every rank runs
  transforms t = rank, rank + N, ... of every image, the float32 results are gathered and the float64 mean is
  formed in the reference's order.

Nothing is exchanged on the data path of a forward pass; per-file results are gathered on rank 0, which logs
the averages.  The PSNR / SSIM values do not depend on N (tests/test_shard.py, tests/test_multi_rank_gpu.py).
"""

import logging
import os
import time

import DCSCN
from dcscn_amd import shard
from helper import args, utilty as util

args.flags.DEFINE_boolean("save_results", True, "Save result, bicubic and loss images.")
args.flags.DEFINE_boolean("compute_bicubic", False, "Compute bicubic performance.")

FLAGS = args.get()


def main(not_parsed_args):
    if len(not_parsed_args) > 1:
        print("Unknown args:%s" % not_parsed_args)
        exit()
    group = shard.init_from_env()
    if group.world > 1:
        FLAGS.gpu_device_id = group.local_rank

    model = DCSCN.SuperResolution(FLAGS, model_name=FLAGS.model_name)
    if FLAGS.frozenInference:
        model.load_graph(FLAGS.frozen_graph_path)             # evaluate.py:50-52 of the reference
        model.build_summary_saver(with_saver=False)
    else:
        model.build_graph()
        model.build_summary_saver()
    model.init_all_variables()

    test_list = ["set5", "set14", "bsd100"] if FLAGS.test_dataset == "all" else [FLAGS.test_dataset]

    for i in range(FLAGS.tests):
        if not FLAGS.frozenInference:
            model.load_model(FLAGS.load_model_name, trial=i, output_log=True if FLAGS.tests > 1 else False)

        if FLAGS.compute_bicubic:
            for test_data in test_list:
                print(test_data)
                evaluate_bicubic(model, test_data, group)

        for test_data in test_list:
            with util.deferred_saves():                        # --save_results: the PNGs of a data set are encoded on worker threads
                evaluate_model(model, test_data, group)
    group.close()


def evaluate_bicubic(model, test_data, group):
    test_filenames = util.get_files_in_directory(FLAGS.data_dir + "/" + test_data)
    mine = [model.evaluate_bicubic(f, print_console=False) for f in group.my_items(test_filenames)]
    results = group.gather(mine)
    if group.rank == 0:
        logging.info("Bicubic Average [%s] PSNR:%f, SSIM:%f" % (
            test_data, sum(r[0] for r in results) / len(results), sum(r[1] for r in results) / len(results)))


def _pixel_counts(filenames):
    from PIL import Image
    sizes = []
    for f in filenames:
        with Image.open(f) as im:
            sizes.append(im.size[0] * im.size[1])
    return sizes


def _evaluate_file(model, filename, save):
    start = time.time()
    if save:
        psnr, ssim = model.do_for_evaluate_with_output(filename, output_directory=FLAGS.output_dir, print_console=False)
    else:
        psnr, ssim = model.do_for_evaluate(filename, print_console=False)
    return psnr, ssim, time.time() - start


def evaluate_model(model, test_data, group):
    test_filenames = util.get_files_in_directory(FLAGS.data_dir + "/" + test_data)
    if shard.split_ensemble(len(test_filenames), group.world, model.self_ensemble):
        # every rank walks every file; inside model.do the transforms of the image are spread over the ranks
        model.ensemble_group = group
        try:
            results = [_evaluate_file(model, f, FLAGS.save_results and group.rank == 0) for f in test_filenames]
        finally:
            model.ensemble_group = None
    else:
        mine = shard.assign_longest_first(_pixel_counts(test_filenames), group.world)[group.rank]
        if not FLAGS.save_results and os.environ.get("DCSCN_EVAL_PIPELINE", "1") != "0":
            # decode of the next files and PSNR / SSIM of the previous ones on worker threads, the device driven from this one
            local = model.do_for_evaluate_many([test_filenames[i] for i in mine])
            tagged = group.gather([(i,) + r for i, r in zip(mine, local)])
        else:
            tagged = group.gather([(i,) + _evaluate_file(model, test_filenames[i], FLAGS.save_results) for i in mine])
        results = [r[1:] for r in sorted(tagged)]             # back to file order
    if group.rank == 0:
        n = len(results)
        logging.info("Model Average [%s] PSNR:%f, SSIM:%f, Time (s): %f" % (
            test_data, sum(r[0] for r in results) / n, sum(r[1] for r in results) / n, sum(r[2] for r in results) / n))
        if os.environ.get("DCSCN_EVAL_DUMP"):               # tests: the per-file values with full precision
            with open(os.environ["DCSCN_EVAL_DUMP"], "a") as f:
                for name, r in zip(test_filenames, results):
                    f.write("%s %s %r %r\n" % (test_data, os.path.basename(name), float(r[0]), float(r[1])))


if __name__ == "__main__":
    args.run(main)
