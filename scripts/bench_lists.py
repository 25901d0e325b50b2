"""Ordered top-k neighbour lists of every row (cx_topk_lists_rows): timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch, cortex_amd, sys
from cortex_amd import _lib
L=_lib.load()
n,d=int(sys.argv[1]),int(sys.argv[2])
gen=torch.empty((n,d),dtype=torch.float32,device="cuda:0")
assert L.cx_synth_fill_dev(0,gen.data_ptr(),20260313,20260313,20260315,n//50,0,n,d,1)==0
ids=np.zeros((n,16),np.uint8); ids[:,8:]=np.arange(n,dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n,8)
h=cortex_amd.HipIndex(d); h.insert_batch_dev(ids,gen.data_ptr(),n,d)
lr,ls,lc=h.topk_lists_rows(100,np.arange(2048,dtype=np.uint32))
for rep in range(2):
    t0=time.perf_counter(); lr,ls,lc=h.topk_lists_rows(100,None); print("all rows top-100: %.4f s" % (time.perf_counter()-t0), int((lc==100).sum()), flush=True)
