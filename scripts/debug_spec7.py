import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np, cases
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from test_gpu_parity import _run_batch
n=5000; m=4; B=32
seq = _run_batch(32 * 512, (0, 0, 0), 2, [n], 4, 100)
PfT = seq[(0,"Pf")]
nb=(n+B-1)//B
true_out = np.array([[PfT[min((b+1)*B,n)-1,0,0], PfT[min((b+1)*B,n)-1,0,1], PfT[min((b+1)*B,n)-1,1,1]] for b in range(nb)])
lib=L.lib()
lib.csr_debug_chain_step.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.c_uint32,C.c_int,C.POINTER(C.c_uint)]
lib.csr_debug_read.argtypes=[C.c_void_p,C.c_int,C.c_void_p,C.c_int64]
mp = ModelParams(state_dim=2)
b = DeviceBatch(0, block_len=B, warm=(0,0,0))
b.configure(mp, m, [n])
data, munc = cases.synth(n, m, 100, mask_frac=0.02, outlier_frac=0.01)
lam, kap, qs = cases.multipliers(n, 100)
b.upload(0, data, munc); b.upload_multipliers(0, lam, kap, qs); b.stats()
flags = L.RETURN_NLL | L.USE_LAMBDA | L.USE_KAPPA | L.USE_QSCALE
def rd(buf):
    a=np.zeros((nb,4),np.float32); lib.csr_debug_read(b._ctx, buf, a.ctypes.data_as(C.c_void_p), a.nbytes); return a
cnt=C.c_uint()
lib.csr_debug_chain_step(b._ctx,0,0,0,flags,0,C.byref(cnt))
cin=rd(0); A=rd(1)
print("after spec: A[0] vs true", A[0,:3], true_out[0]); print("cin[0..2]", cin[:3])
which=0
for it in range(8):
    lib.csr_debug_chain_step(b._ctx,0,1,which,flags,0,C.byref(cnt))
    cin=rd(0); A=rd(1); Bb=rd(2)
    new = Bb if which==0 else A
    nwrong = int(np.sum(np.any(new[:,:3]!=true_out,axis=1)))
    first = np.nonzero(np.any(new[:,:3]!=true_out,axis=1))[0][:8]
    cinwrong = np.nonzero(np.any(cin[1:,:3]!=true_out[:-1],axis=1))[0][:8]+1
    print("iter",it,"reruns",cnt.value,"out wrong",nwrong,"first",first,"cin wrong first",cinwrong)
    if it==0:
        print("  new[1]",new[1],"true",true_out[1],"cin[1]",cin[1],"A[0]",A[0])
    which^=1
    if cnt.value==0: break
print("---- bit-level")
b2 = DeviceBatch(0, block_len=B, warm=(0,0,0))
b2.configure(mp, m, [n]); b2.upload(0, data, munc); b2.upload_multipliers(0, lam, kap, qs); b2.stats()
lib.csr_debug_chain_step(b2._ctx,0,0,0,flags,0,C.byref(cnt))
lib.csr_debug_chain_step(b2._ctx,0,1,0,flags,0,C.byref(cnt))
TN = ((nb+63)//64)*B*64
tpf = np.zeros((TN,4),np.float32); lib.csr_debug_read(b2._ctx,3,tpf.ctypes.data_as(C.c_void_p),tpf.nbytes)
def slot(bb,s): return ((bb>>6)*B+s)*64+(bb&63)
for s in range(0,6):
    r=32+s
    got=tpf[slot(1,s)]; exp=PfT[r].ravel()
    print("row",r,"got",got.view(np.uint32), "exp",exp.view(np.uint32), "ulp diff", got.view(np.int32).astype(np.int64)-exp.view(np.int32).astype(np.int64))
