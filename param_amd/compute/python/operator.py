"""Operator plug-in contract -- own restatement of reference ``train/compute/python/lib/operator.py:8-67``:
``build / cleanup / forward / create_grad / backward``; an INSTANCE is registered by name; a duplicate
name raises ``ValueError``."""
from __future__ import annotations

import abc
import logging

logger = logging.getLogger(__name__)


class OperatorInterface(metaclass=abc.ABCMeta):
    def __init__(self):
        self.device = None  # set by the config loader ("cuda", "cuda:0", "cpu")

    def build(self, *args, **kwargs):
        """construct and initialise the operator"""

    def cleanup(self):
        """reset state, release resources"""

    @abc.abstractmethod
    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def create_grad(self):
        raise NotImplementedError

    def backward(self):
        raise NotImplementedError


op_map: dict = {}


def register_operator(name: str, operator: OperatorInterface) -> None:
    logger.debug(f"register op: {name}")
    if name in op_map:
        raise ValueError(f"Duplicate operator registration name: {name}")
    op_map[name] = operator


def register_operators(op_dict: dict) -> None:
    for name, operator in op_dict.items():
        register_operator(name, operator)
