import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E
L=10000000
eng=E.Engine(0); eng.set_update_mode("gemm")
ds=E.Dataset(eng, generate=(3456,256,512,0,L))
lo,hi,cnt=E.column_minmax(ds)
init=E.randinit_from_bbox(lo,hi,cnt,256,256,7)
cb=E.Codebook(eng,init,E.TOPOL_HEXA,E.NEIGH_BUBBLE,256,256)
pos=0
for frac in (0.0,0.05,0.1,0.2,0.3,0.4,0.5,0.7,0.9):
    target=int(frac*L)//4096*4096
    if target>pos:
        E.som_train(cb,ds,L,0.05,128.0,batch=4096,start_iter=pos,count=target-pos,data_first=pos,trace=False); pos=target
    s0=eng.scan_stats()
    eng.timing(True); eng.timing_reset()
    E.som_train(cb,ds,L,0.05,128.0,batch=4096,start_iter=pos,count=4096,data_first=pos,trace=False); pos+=4096
    eng.timing(False)
    s1=eng.scan_stats(); t=eng.timing_table()
    print("at %4.0f%%: level-2 pairs/sample %.1f, rerank groups/sample %.2f rows %.1f; k_dist_l2 %.0f us, L1 %.0f us, update %.0f us"%(100*frac,(s1["l2_pairs"]-s0["l2_pairs"])/4096,(s1["groups"]-s0["groups"])/4096,(s1["rows"]-s0["rows"])/4096,1e3*t["k_dist_l2"][1],1e3*t["k_dist_mfma_bf16"][1],1e3*t["k_som_update_gemm"][1]),flush=True)
