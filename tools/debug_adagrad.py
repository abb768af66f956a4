import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.embbag_oracle import COracle
from param_amd import BatchedEmbeddingBagMI355
DEV="cuda:0"; orc=COracle(); rng=np.random.default_rng(8)
for D in (128, 56):
    rows,B,L=[4000,700],300,10
    m=BatchedEmbeddingBagMI355(rows,D,device=DEV,init="normal",seed=D,learning_rate=0.05,optimizer="rowwise_adagrad",eps=1e-6)
    W=[m.table(t).cpu().numpy().copy() for t in range(2)]
    idx=torch.cat([torch.randint(0,r,(B*L,)) for r in rows]); idx[:450]=3
    off=torch.arange(2*B+1)*L
    grad=torch.from_numpy(rng.standard_normal((B,2*D)).astype(np.float32))
    m.adagrad_step_(grad.to(DEV), idx.to(DEV), off.to(DEV))
    torch.cuda.synchronize()
    for t in range(2):
        s,e=t*B*L,(t+1)*B*L
        g=np.ascontiguousarray(grad.numpy()[:,t*D:(t+1)*D])
        Wo=W[t].copy(); mo=np.zeros(rows[t],np.float32)
        orc.bwd_rowwise_adagrad(Wo,mo,idx.numpy()[s:e],np.arange(B)*L,g,lr=0.05,eps=1e-6)
        gw=m.table(t).cpu().numpy(); gm=m.momentum_table(t).cpu().numpy()
        touched=np.bincount(idx.numpy()[s:e],minlength=rows[t])>0
        badw=np.nonzero((gw!=W[t]).any(1) & ~touched)[0]; badm=np.nonzero((gm!=0)&~touched)[0]
        print(f"D={D} t={t} untouched-but-changed W rows {badw[:10]} ({len(badw)}), mom rows {badm[:10]} ({len(badm)})")
        dm=np.abs(gm-mo); dw=np.abs(gw-Wo).max(1)
        print("   max |mom diff|", dm.max(), "at", dm.argmax(), "cnt", np.bincount(idx.numpy()[s:e],minlength=rows[t])[dm.argmax()], " max |W diff|", dw.max(), "at", dw.argmax())
        wrong=np.nonzero(dm>1e-4*(np.abs(mo)+1e-6))[0]; print("   rows with wrong mom:", wrong[:12], len(wrong), "their counts", np.bincount(idx.numpy()[s:e],minlength=rows[t])[wrong[:12]])
