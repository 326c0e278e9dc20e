import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import numpy as np
from common import make_pair
orc, eng = make_pair(1,1,3,7)
A = eng.create_mat()
rp,ci,val = A.host()
Ao = orc.create_mat()
print(A.nbrows, A.nblocks, Ao.nrows, Ao.nnz)
print(rp); print(Ao.rowptr)
print(ci[:20]); print(Ao.colidx[:20])
print(A.layout())
