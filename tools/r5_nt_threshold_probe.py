import json, os, sys, torch
sys.path.insert(0, "/root/repo")
import param_amd
from param_amd.embedding_bag import _TableSet, _fwd
from param_amd.indices import tbe_request
dev = torch.device("cuda", 0)
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", layout="tbd", seed=1000, fused_update=False)
req = {"zipf": tbe_request([R] * T, B, [L] * T, alpha=1.05, device=dev, seed=1), "uniform": tbe_request([R] * T, B, [L] * T, alpha=0.0, device=dev, seed=2)}
ts = _TableSet([m.table(t) for t in range(T)], "tbd")
out = torch.empty((T, B, D), device=dev)
for rnd in range(2):
    for thr in (0, 1, 512, 2048, 8192, 32768, 131072, 1 << 20):
        param_amd.set_tuning(nt_loads=thr if thr else -1)
        for dist in ("zipf", "uniform"):
            i, o = req[dist]
            for _ in range(10): _fwd(ts, i, o, B, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(40): _fwd(ts, i, o, B, out=out)
            e1.record(); torch.cuda.synchronize()
            print(json.dumps({"round": rnd, "nt_threshold": thr, "indices": dist, "us": round(e0.elapsed_time(e1) * 1e3 / 40, 2)}), flush=True)
