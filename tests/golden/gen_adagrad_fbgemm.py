#!/usr/bin/env python3
"""Pin the row-wise Adagrad oracle to fbgemm_gpu's OWN outputs -- the moment an image has it.

The reference configures ``OptimType.EXACT_ROWWISE_ADAGRAD`` for its TBE operators (train/comms/pt/comms_utils.py:2014,
train/compute/python/workloads/pytorch/split_table_batched_embeddings_ops.py:279-300); fbgemm_gpu is an unpinned third-party
dependency that is not installable in the build image (no index access), so ``oracle_embbag_bwd_rowwise_adagrad_wd_f32``
restates its published formulas and says PARITY UNPINNED (oracle/embbag_oracle.c).  This script closes that gap where it
can run: with ``fbgemm_gpu`` importable and a GPU visible it drives fbgemm's ``SplitTableBatchedEmbeddingBagsCodegen``
(fp32 weights, SUM pooling, EXACT_ROWWISE_ADAGRAD, weight decay NONE / L2 / DECOUPLE) over small seeded requests and writes
``tests/golden/adagrad_fbgemm.npz`` = inputs + fbgemm's updated weights and optimizer state; ``tests/test_oracle.py::
test_oracle_rowwise_adagrad_pinned_to_fbgemm_fixture`` then holds the oracle (and, on a GPU, the HIP kernel) against it.
Without fbgemm_gpu it says so and exits 0: nothing is fabricated.

    python tests/golden/gen_adagrad_fbgemm.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    try:
        import torch
        from fbgemm_gpu.split_embedding_configs import EmbOptimType as OptimType
        from fbgemm_gpu.split_table_batched_embeddings_ops_common import EmbeddingLocation, PoolingMode
        from fbgemm_gpu.split_table_batched_embeddings_ops_training import (ComputeDevice, SplitTableBatchedEmbeddingBagsCodegen,
                                                                          WeightDecayMode)
    except Exception as exc:   # ImportError, or a missing shared object behind it
        print(f"gen_adagrad_fbgemm: fbgemm_gpu is not importable here ({exc.__class__.__name__}: {exc}); no fixture written -- "
              "the row-wise Adagrad oracle stays PARITY UNPINNED (oracle/embbag_oracle.c)")
        return 0
    if not torch.cuda.is_available():
        print("gen_adagrad_fbgemm: fbgemm_gpu imports but no GPU is visible (its TBE training op is a device op); no fixture written")
        return 0
    import fbgemm_gpu

    dev = torch.device("cuda:0")
    out = {"fbgemm_gpu_version": np.array(getattr(fbgemm_gpu, "__version__", "unknown"))}
    rng = np.random.default_rng(0)
    cases = [("none", WeightDecayMode.NONE, 0.0), ("l2", WeightDecayMode.L2, 0.01), ("decouple", WeightDecayMode.DECOUPLE, 0.01)]
    for tag, wdm, wd in cases:
        rows, D, B, L, lr, eps = [50, 7, 300], 16, 12, 5, 0.05, 1e-5
        T = len(rows)
        op = SplitTableBatchedEmbeddingBagsCodegen(
            [(r, D, EmbeddingLocation.DEVICE, ComputeDevice.CUDA) for r in rows], optimizer=OptimType.EXACT_ROWWISE_ADAGRAD,
            learning_rate=lr, eps=eps, weight_decay=wd, weight_decay_mode=wdm, pooling_mode=PoolingMode.SUM,
            stochastic_rounding=False, device=dev)
        W0 = [rng.standard_normal((r, D)).astype(np.float32) for r in rows]
        for t, w in enumerate(op.split_embedding_weights()):
            w.data.copy_(torch.from_numpy(W0[t]))
        idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
        idx[:7] = 3                                     # duplicates inside and across bags of table 0
        off = (np.arange(T * B + 1) * L).astype(np.int64)
        grads = []
        for step in range(2):                           # two steps: the state carries over
            g = rng.standard_normal((B, T * D)).astype(np.float32)
            grads.append(g)
            y = op(torch.from_numpy(idx).to(dev), torch.from_numpy(off).to(dev))
            y.backward(torch.from_numpy(g).to(dev))
        torch.cuda.synchronize()
        state = op.split_optimizer_states()
        out.update({f"{tag}.rows": np.array(rows), f"{tag}.hp": np.array([D, B, L, lr, eps, wd], dtype=np.float64), f"{tag}.idx": idx,
                    f"{tag}.off": off, f"{tag}.grads": np.stack(grads)})
        for t in range(T):
            out[f"{tag}.W0.{t}"] = W0[t]
            out[f"{tag}.W.{t}"] = op.split_embedding_weights()[t].detach().cpu().numpy()
            st = state[t][0] if isinstance(state[t], (tuple, list)) else state[t]
            out[f"{tag}.mom.{t}"] = st.detach().cpu().numpy().reshape(-1)
    path = os.path.join(HERE, "adagrad_fbgemm.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} (fbgemm_gpu {out['fbgemm_gpu_version']})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
