"""Super-resolve one image file -- same command line as the reference's sr.py:

    python sr.py --file=your_image.png [--scale=3] [--layers=8 --filters=96] ...

Results go to output/<model name>/ (original, bicubic, Y and colour results).  The model flags must
match the checkpoint, exactly as with the reference (sr.py:1-27).
"""

import DCSCN
from helper import args

args.flags.DEFINE_string("file", "image.jpg", "Target filename")
FLAGS = args.get()


def main(_):
    model = DCSCN.SuperResolution(FLAGS, model_name=FLAGS.model_name)
    model.build_graph()
    model.build_optimizer()
    model.build_summary_saver()

    model.init_all_variables()
    model.load_model()

    model.do_for_file(FLAGS.file, FLAGS.output_dir)


if __name__ == "__main__":
    args.run(main)
