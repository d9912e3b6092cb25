set -e
R=$GRAFT_REPO_ROOT
T=$(mktemp -d)
mkdir -p $T/data $T/models
cp -r $R/tests/golden/set5 $T/data/set5
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
PY
cd $T
# L2 checkpoint shipped in fixtures (legacy topology) -> full CLI path incl. file output
time python $R/evaluate.py --test_dataset=set5 --layers=2 --filters=4 --min_filters=4 --use_nin=false --reconstruct_filters=4 --self_ensemble=8 --checkpoint_dir=$R/tests/golden/models --data_dir=$T/data --output_dir=$T/out --log_filename=$T/log.txt 2>&1 | tail -3
time python $R/evaluate.py --test_dataset=set5 --layers=2 --filters=4 --min_filters=4 --use_nin=false --reconstruct_filters=4 --self_ensemble=8 --nosave_results --checkpoint_dir=$R/tests/golden/models --data_dir=$T/data --output_dir=$T/out --log_filename=$T/log.txt 2>&1 | tail -2
python - <<PY
import sys, time, os, cProfile, pstats
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np
from test_host import _flags
from dcscn_amd.model import SuperResolution
L7 = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8, reconstruct_layers=0, pixel_shuffler_filters=1)
m = SuperResolution(_flags(checkpoint_dir="$T/models", self_ensemble=8, **L7))
m.build_graph(); m.init_all_variables()
m.load_weights(dict(np.load("$R/tests/golden/weights_L7_x2.npz")))
files = sorted(os.listdir("$T/data/set5"))
m.do_for_evaluate("$T/data/set5/" + files[0])
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
for f in files: m.do_for_evaluate("$T/data/set5/" + f)
dt = time.time() - t0
pr.disable()
print("L7 x2 ens8 Set5: %.3f s per image" % (dt / len(files)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
PY
