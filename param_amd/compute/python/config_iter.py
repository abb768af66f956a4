"""Range expansion of benchmark configs -- the slice of reference ``train/compute/python/lib/iterator.py``
(``RangeConfigIterator`` ``:179-247``, ``create_range_iter`` ``:59-176``, ``DefaultConfigIterator`` ``:250-270``) and of the
operator's own input iterator (``workloads/pytorch/split_table_batched_embeddings_ops.py:33-84``) that the batched
EmbeddingBag operator's configs use.

An argument ``{"type": T, "value": V, "__range__": ["value"]}`` expands to
  int / long        ``V = [first, last(, step)]``: every ``first <= x <= last`` by ``step`` (inclusive, ``generator.py:4-8``)
  str / bool / device   ``V`` = list of alternatives
  genericlist / tuple   ``V`` = list of typed items, each expanded the same way: the Cartesian product of the items
and a config is the Cartesian product of its arguments, FIRST argument outermost, positional arguments outside keyword
arguments (``ListProduct`` / ``TableProduct``, ``generator.py:29-178``).  Ids are ``"<variant>_<n>"``.  float / none
arguments never range (the reference does not support it either).  Pinned by tests/golden/range_configs.json, produced
by the reference's ``RangeConfigIterator`` (tests/golden/gen_range_configs.py).

The operator's input iterator reads ``[batch_size, pooling_factor]`` with ``__range__`` (inclusive range) or ``__list__``
(alternatives) on either; that class cannot be imported without fbgemm_gpu, so this part is a restatement only.
"""
from __future__ import annotations

import copy
import itertools
from typing import Any, Dict, Iterator, List, Tuple

ATTR_COPY, ATTR_RANGE, ATTR_LIST = "__copy__", "__range__", "__list__"
META_ATTRS = (ATTR_COPY, ATTR_RANGE, ATTR_LIST)


def full_range(a: int, b: int, s: int = 1) -> range:
    """inclusive range: a <= x <= b by step s"""
    return range(a, b + 1, s)


def _alternatives(arg: Dict[str, Any]) -> List[Dict[str, Any]]:
    """every concrete form of one typed argument, in the reference's iteration order"""
    ranged = ATTR_RANGE in arg and "value" in set(arg[ATTR_RANGE])
    kind = arg.get("type")
    if not ranged or kind in ("float", "double", "none"):
        return [arg]
    if kind in ("int", "long"):
        values = list(full_range(*arg["value"]))
    elif kind in ("str", "bool", "device"):
        values = list(arg["value"])
    elif kind in ("genericlist", "tuple"):
        per_item = [_alternatives(item) for item in arg["value"]]
        values = [list(combo) for combo in itertools.product(*per_item)]
    else:
        raise ValueError(f"__range__ on an argument of type {kind!r} is not supported by this build")
    out = []
    for v in values:
        a = dict(arg)
        a["value"] = copy.deepcopy(v)
        out.append(a)
    return out


def _strip(arg: Dict[str, Any]) -> Dict[str, Any]:
    a = {k: v for k, v in arg.items() if k not in META_ATTRS}
    if a.get("type") in ("genericlist", "tuple") and isinstance(a.get("value"), list):
        a["value"] = [_strip(x) if isinstance(x, dict) else x for x in a["value"]]
    return a


def range_config_iterator(variants: List[Dict[str, Any]]) -> Iterator[Tuple[str, Dict[str, Any]]]:
    """``RangeConfigIterator``: yields ``(id, {"args": [...], "kwargs": {...}})`` with meta attributes removed from
    the positional arguments (the reference leaves them on keyword arguments; values are what matters)."""
    for var_id, config in enumerate(variants):
        args = config.get("args") or []
        kwargs = config.get("kwargs") or {}
        arg_alts = [_alternatives(a) for a in args]
        kw_keys = list(kwargs)
        kw_alts = [_alternatives(kwargs[k]) for k in kw_keys]
        for config_id, combo in enumerate(itertools.product(*arg_alts, *kw_alts)):
            yield (f"{var_id}_{config_id}",
                   {"args": [_strip(a) for a in combo[:len(args)]],
                    "kwargs": {k: _strip(a) for k, a in zip(kw_keys, combo[len(args):])}})


def default_config_iterator(variants: List[Dict[str, Any]]) -> Iterator[Tuple[int, Dict[str, Any]]]:
    """``DefaultConfigIterator``: the variants as written"""
    for i, config in enumerate(variants):
        yield i, config


def tbe_input_iterator(inputs: List[Dict[str, Any]]) -> Iterator[Tuple[str, List[Any]]]:
    """``SplitTableBatchedEmbeddingBagsCodegenInputIterator``: ``(id, [batch_size, pooling_factor])``; every id starts
    with variant 0 because the reference never advances its variant counter (``:52-81``)"""
    for inp in inputs:
        alts = []
        for arg in inp["args"]:
            if ATTR_RANGE in arg:
                alts.append(list(full_range(*arg["value"])))
            elif ATTR_LIST in arg:
                alts.append(list(arg["value"]))
            else:
                alts.append([arg["value"]])
        for config_id, combo in enumerate(itertools.product(*alts)):
            yield f"0_{config_id}", list(combo)


BUILD_ITERATORS = {"RangeConfigIterator": range_config_iterator, "DefaultConfigIterator": default_config_iterator, None: default_config_iterator}
