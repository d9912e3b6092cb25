"""Drop-in ``helper`` package: the module names the reference's CLIs import (helper/args.py,
helper/utilty.py, helper/loader.py, helper/tf_graph.py), re-hosted without TensorFlow."""
