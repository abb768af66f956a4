"""Canned EmbeddingBag shapes of the reference driver (``driver.py emb -d {A,B}``).

Values are the reference's benchmark configuration data
(train/compute/pt/dataset.py:56-82): tuples ``(features, embdim, nnz, batch)``.
Only the ``emb`` datasets exist here: gemm / mlp shapes belong to kernels outside the
hot path.
"""

_BATCHES_A = [512 << i for i in range(8)]  # 512 .. 65536
emb_A = [(14000000, 128, 30, b) for b in _BATCHES_A] + [(26000000, 128, 30, b) for b in _BATCHES_A]
emb_B = [(4800000, 56, 34, 2048 << i) for i in range(6)]  # 2048 .. 65536

# BASELINE.json configs[1]/[2]: (tables, rows, dim, nnz, batch)
emb_mi355_fp32 = (48, 10000000, 128, 20, 8192)   # largest table count that fits 288 GB in fp32
emb_mi355_bf16 = (64, 10000000, 128, 20, 8192)
