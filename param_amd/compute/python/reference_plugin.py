"""Registers the MI355X batched EmbeddingBag operator in the REFERENCE's own ``train/compute/python`` registries, so its
unmodified ``pytorch/run_benchmark.py`` drives the HIP kernels from its own JSON configs.

The reference resolves three names from a config (``examples/pytorch/configs/split_table_batched_embeddings_ops.json``):
the operator (``lib/operator.py:48-54`` ``register_operator``), its input iterator (``lib/iterator.py:281``
``register_config_iterator``) and its input data generator (``lib/data.py:27`` ``register_data_generator``).  In the
reference all three live in ``workloads/pytorch/split_table_batched_embeddings_ops.py``, which imports ``fbgemm_gpu`` at
module level; this module supplies the same three names WITHOUT that import, on top of the reference's base classes, so
``make_op_config`` (``lib/config.py:70-107``) finds a complete entry.

Needs the reference package importable (default ``param_bench.train.compute.python``); nothing here is used by this
build's own runner (``run_benchmark.py``), which has its own registry.
"""
from __future__ import annotations

import copy
import importlib

OP_NAME = "SplitTableBatchedEmbeddingBagsCodegen"
ITERATOR_NAME = "SplitTableBatchedEmbeddingBagsCodegenInputIterator"
GENERATOR_NAME = "SplitTableBatchedEmbeddingBagsCodegenInputDataGenerator"


def register_in_reference(package: str = "param_bench.train.compute.python", replace: bool = False) -> dict:
    """Returns ``{"operator": ..., "iterator": ..., "generator": ...}`` (the registered objects).  ``replace``: overwrite
    entries the reference's own fbgemm-backed module already registered (fbgemm_gpu installed) instead of raising."""
    ref_operator = importlib.import_module(package + ".lib.operator")
    ref_iterator = importlib.import_module(package + ".lib.iterator")
    ref_data = importlib.import_module(package + ".lib.data")

    from .config_iter import tbe_input_iterator
    from .split_table_batched_embeddings_ops import SplitTableBatchedEmbeddingBagsCodegenOp, generate_batched_request

    class InputIterator(ref_iterator.ConfigIterator):
        """reference ``:33-84``: (id, {"args": [num_tables, rows, dim, batch_size, pooling_factor, weighted,
        weights_precision] as typed argument dicts, "kwargs": {}}) for every [batch_size, pooling_factor] combination"""

        def __init__(self, configs, key, device):
            super().__init__(configs, key, device)
            b = configs["build"]["args"]
            self._fixed = (b[0], b[1], b[2], b[4], b[5])
            self._gen = self._generate()

        def _generate(self):
            nt, rows, dim, weighted, prec = self._fixed
            for inp in self.configs[self.key]:
                protos = [{k: v for k, v in a.items() if not k.startswith("__")} for a in inp["args"]]
                for cid, (batch, pooling) in tbe_input_iterator([inp]):
                    bs, pf = dict(protos[0], value=batch), dict(protos[1], value=pooling)
                    yield cid, copy.deepcopy({"args": [nt, rows, dim, bs, pf, weighted, prec], "kwargs": {}})

        def __next__(self):
            return next(self._gen)

    class InputDataGenerator(ref_data.DataGenerator):
        """reference ``:138-229``: ([indices, offsets, per_sample_weights], {}) on ``device`` in the TBE layout"""

        def get_data(self, config, device, alpha=1):
            a = config["args"]
            num_tables = a[0]["value"]
            return ([*generate_batched_request(num_tables, a[1]["value"], a[3]["value"], a[4]["value"], float(alpha),
                                               a[5]["value"], device)], {})

    op = SplitTableBatchedEmbeddingBagsCodegenOp()
    for table, name, obj in ((ref_operator.op_map, OP_NAME, op),
                             (ref_iterator.config_iterator_map, ITERATOR_NAME, InputIterator),
                             (ref_data.data_generator_map, GENERATOR_NAME, InputDataGenerator)):
        if name in table and not replace:
            raise ValueError(f"{name} is already registered in the reference (pass replace=True to take it over)")
        table[name] = obj
    return {"operator": op, "iterator": InputIterator, "generator": InputDataGenerator}
