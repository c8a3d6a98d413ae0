#!/usr/bin/env python3
"""CPU-side fuzz of the lower-MAC lane code (csrc/lmac_core.hpp built for the host, tests/emul) against the REFERENCE build
(oracle/_ref/libtetra_lmac_ref.so; container only): clean / noisy / burst-error / erasure / random / constant / periodic rows
of every coded block kind, decoded bits and CRC verdicts equal.  python profiles/fuzz_lmac_cpu.py <seed> <seconds>
Test infrastructure (uses oracle/ as the checker)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_binding as ref
from tests.emul import lmac_emul_bind
CODED=(0,1,2,4,5)
STRIDE=436
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
t0=time.time(); total=0; bad=[]
while time.time()-t0 < float(sys.argv[2] if len(sys.argv)>2 else 60):
    t=int(rng.choice(CODED))
    n345,n2,n1,a,_=ref.BLK_PARAM[t]
    n=64
    si=rng.integers(0,2**32,n,dtype=np.uint64).astype(np.uint32)
    if t==ref.TPSAP_T_SB1: si[:]=ref.SCRAMB_INIT
    rows=rng.integers(0,256,(n,STRIDE),dtype=np.uint8)
    for b in range(n):
        mode=int(rng.integers(0,8))
        if mode<=3:
            sent=rng.integers(0,2,n1).astype(np.uint8)
            r=ref.lmac_encode(t,sent,si[b]).astype(np.uint8)
            if mode==1: r=r^(rng.random(n345)<rng.uniform(0,0.25))
            if mode==2:   # burst errors
                s=int(rng.integers(0,n345)); l=int(rng.integers(1,60)); r[s:s+l]^=1
            if mode==3:   # erasures (other soft class) sprinkled
                m=rng.random(n345)<rng.uniform(0,0.5); r=np.where(m, rng.choice(np.array([2,7,0x80,0x7f],np.uint8),n345), r)
            rows[b,:n345]=r
        elif mode==4: rows[b,:n345]=rng.integers(0,2,n345)
        elif mode==5: rows[b,:n345]=rng.choice(np.array([0,1,0xff,0xfe,2,7],np.uint8),n345)
        elif mode==6: rows[b,:n345]=int(rng.choice([0,1,2,0xff]))          # constant rows: ties everywhere
        else: rows[b,:n345]=np.tile(rng.integers(0,3,int(rng.integers(1,9))).astype(np.uint8), n345)[:n345]   # short periodic patterns
    want=np.zeros((n,n2),np.uint8); wok=np.zeros(n,np.int32)
    for b in range(n): want[b],wok[b]=ref.lmac_decode(t,rows[b],si[b])
    got,gok=lmac_emul_bind.decode_batch(t,rows,si)
    total+=n
    if not (np.array_equal(got,want) and np.array_equal(gok,wok)):
        bad.append((t,int(np.flatnonzero((got!=want).any(axis=1)|(gok!=wok))[0])))
        if len(bad)>5: break
print(dict(blocks=total,bad=bad,seconds=round(time.time()-t0,1)))
