"""Mirror of the reference's train/compute/python operator plug-in contract for the batched
EmbeddingBag operator (lib/operator.py:8-67, workloads/pytorch/split_table_batched_embeddings_ops.py)."""
from .operator import OperatorInterface, op_map, register_operator  # noqa: F401
from . import split_table_batched_embeddings_ops  # noqa: F401  (registers the operator)
