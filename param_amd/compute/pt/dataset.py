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

# BASELINE.json configs[4]: MLPerf DLRM-v2 (Criteo 1TB, multi-hot) embedding tables -- model constants of the
# public MLCommons DLRM-v2 benchmark (NOT in the reference repo; SURVEY.md section 8d marks them "memory, verify"):
# 26 tables, embedding dim 128, rows and multi-hot pooling sizes per table.
criteo_v2_rows = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209,
                  11938, 155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]
criteo_v2_multi_hot = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
criteo_v2_dim = 128


def criteo_v2_mixed_dims(rows=None):
    """Embedding dims of a MIXED-dim DLRM over the Criteo tables (BASELINE configs[4] says "mixed-dim"; the reference's hook is
    train/comms/pt/dlrm.py:384-385 ``mixed_dim -> torch.cat(ly, dim=1)``, dims from :506-557): by table size -- rows >= 10 M -> 128,
    >= 100 K -> 64, >= 1 K -> 32, else 16 (the round-5 review's rule; the reference publishes none)."""
    rows = criteo_v2_rows if rows is None else rows
    return [128 if r >= 10_000_000 else 64 if r >= 100_000 else 32 if r >= 1_000 else 16 for r in rows]
