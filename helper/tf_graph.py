"""``helper.tf_graph`` of the reference defined ``TensorflowGraph``, the TF-session base class of the
model.  The MI355X build has no TF graph; the name is kept so ``tf_graph.TensorflowGraph`` still
resolves to the model's base type for code that checks it."""

import dcscn_amd  # noqa: F401
from dcscn_amd.model import SuperResolution as TensorflowGraph          # noqa: F401
