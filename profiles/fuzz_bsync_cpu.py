#!/usr/bin/env python3
"""CPU-side fuzz of the burst synchroniser: the reference's own tetra_burst_sync_in / tetra_burst_rx_cb run (oracle/_ref; container
only) == the restatement (oracle/burst_sync_oracle.c) == the kernel logic built for the host (csrc/bsync_core.hpp, tests/emul) on
random adversarial streams (tests/test_burst_sync.py make_stream + bit flips), random call / chunk sizes.
python profiles/fuzz_bsync_cpu.py <seed> <seconds>.  Test infrastructure (uses oracle/ as the checker)."""
import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from oracle import ref_binding as ref, binding as oracle
import test_burst_sync as T
from tests.emul import bsync_emul_bind
seed0=int(sys.argv[1]); secs=float(sys.argv[2])
rng=np.random.default_rng(seed0)
t0=time.time(); streams=0; calls=0
while time.time()-t0<secs:
    seed=int(rng.integers(1000,10**9))
    tx=T.make_stream(ref, seed)
    # extra adversity: random bit flips + random slips
    if rng.integers(0,2):
        m=rng.random(tx.size)<rng.uniform(0,0.02); tx=tx^m.astype(np.uint8)
    r,o,e=ref.ReferenceBurstSync(), oracle.BurstSyncOracle(), bsync_emul_bind.Emul()
    pos=0
    while pos<tx.size:
        n=int(rng.choice([1,37,510,3000,9000])); chunk=int(rng.choice([1,1,2,7,100,509,510]))
        blk=tx[pos:pos+n]; pos+=n
        got=r.feed(blk,chunk); fo=o.feed(blk,chunk)
        want=T._expected_tp_sap_calls(oracle, ref, *fo)
        assert T._same_calls(got,want),(seed,pos,chunk)
        assert r.state==o.state,(seed,pos,chunk)
        calls+=len(got)
    r.close()
    # kernel logic (host build) vs literal machine fed one bit per call
    o2=oracle.BurstSyncOracle(); pos=0
    while pos<tx.size:
        n=int(rng.choice([1,7,37,510,1000,5000,36000])); c=tx[pos:pos+n]; pos+=n
        fo=o2.feed(c,1); fe=e.feed(c)
        for a,b in zip(fo,fe):
            assert np.array_equal(np.asarray(a),np.asarray(b)),(seed,pos,'emul')
        assert o2.state==e.state,(seed,pos,'emul state')
    streams+=1
print(dict(streams=streams,tp_sap_calls=calls,seconds=round(time.time()-t0,1)))
