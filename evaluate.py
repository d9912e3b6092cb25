"""Evaluate a trained model on a dataset -- same command line as the reference's evaluate.py:

    python evaluate.py --test_dataset=set5 [--scale=3] [--save_results=false] ...

Builds LR inputs from data/<test_dataset>/, super-resolves them on the MI355X and logs
``Model Average [<set>] PSNR:..., SSIM:..., Time (s): ...`` (evaluate.py:106-107).

One process drives one GPU (``--gpu_device_id``).  Launched under ``torch.distributed.run`` with N
ranks, the image list is sharded across ranks (contiguous shards, no data-path collective); per-file
results are gathered on rank 0, which logs the averages.
"""

import logging
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
            evaluate_model(model, test_data, group)
    group.close()


def evaluate_bicubic(model, test_data, group):
    test_filenames = util.get_files_in_directory(FLAGS.data_dir + "/" + test_data)
    mine = [model.evaluate_bicubic(f, print_console=False) for f in group.my_items(test_filenames)]
    results = group.gather(mine)
    if group.rank == 0:
        logging.info("Bicubic Average [%s] PSNR:%f, SSIM:%f" % (
            test_data, sum(r[0] for r in results) / len(results), sum(r[1] for r in results) / len(results)))


def evaluate_model(model, test_data, group):
    test_filenames = util.get_files_in_directory(FLAGS.data_dir + "/" + test_data)
    mine = []
    for filename in group.my_items(test_filenames):
        start = time.time()
        if FLAGS.save_results:
            psnr, ssim = model.do_for_evaluate_with_output(filename, output_directory=FLAGS.output_dir,
                                                           print_console=False)
        else:
            psnr, ssim = model.do_for_evaluate(filename, print_console=False)
        mine.append((psnr, ssim, time.time() - start))
    results = group.gather(mine)
    if group.rank == 0:
        n = len(results)
        logging.info("Model Average [%s] PSNR:%f, SSIM:%f, Time (s): %f" % (
            test_data, sum(r[0] for r in results) / n, sum(r[1] for r in results) / n, sum(r[2] for r in results) / n))


if __name__ == "__main__":
    args.run(main)
