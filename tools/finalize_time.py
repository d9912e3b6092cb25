import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import dcscn_oracle as O
from conftest import CONFIGS
from dcscn_amd import engine
for name in ("L12_F196to48_x4", "L7_F32to8_x4", "L7_F32to8_x3", "L7_F32to8_x4_DS"):
    cfg = O.make_config(**CONFIGS[name]); w = O.synthetic_weights(cfg, seed=0)
    for whole in (1, 0):
        eng = engine.Engine(cfg, device=0); eng.set_option("fold_whole_tail", whole)
        for n, _ in eng.tensor_specs(): eng.set_tensor(n, w[n])
        t0 = time.perf_counter(); eng.finalize(); dt = time.perf_counter() - t0
        print("%s fold_whole_tail %d: dcscn_finalize %.3f s" % (name, whole, dt)); eng.close()
