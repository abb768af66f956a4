"""param_amd -- MI355X-native (gfx950) embedding-lookup + DLRM all-to-all hot path of PARAM.

Layout (only what the hot path needs):
  csrc/              hand-written HIP kernels + the C ABI (include/param_amd.h) -> libparam_amd.so
  _lib.py            ctypes binding (no fallback: raises if the library is missing)
  embedding_bag.py   EmbeddingBagMI355 / BatchedEmbeddingBagMI355 (nn.EmbeddingBag / TBE surface)
  indices.py         uniform / Zipf index generation (reference init_indices + a scalable form)
  compute/pt/        mirror of the reference train/compute/pt emb driver CLI
  comms/pt/          mirror of the reference train/comms/pt backend plug-in, metrics and drivers
"""
from ._lib import LIB_PATH, ParamAmdError, load as load_library, set_backward_tuning, set_forward_tuning, set_hybrid_min_tiles, set_hybrid_rest, set_hybrid_tuning, set_sort_tuning, set_tuning  # noqa: F401
from .embedding_bag import (  # noqa: F401
    BatchedEmbeddingBagMI355,
    EmbeddingBagMI355,
    check_request,
    fill_random_,
)

__version__ = "0.1.0"
