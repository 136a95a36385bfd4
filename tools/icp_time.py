import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng=np.random.default_rng(0)
M,_=synth.make_model(rng,5000); M=M.astype(np.float32)
R=synth._rot_axis_angle([0.2,0.5,-0.4],0.8); t=np.array([0.1,0.0,0.7])
S=(M[rng.choice(5000,2500,replace=False)]@R.T+t).astype(np.float32)
Tinv=np.linalg.inv(synth._se3(R,t))
sc=LcpScorer()
for n in (1,8,64,256):
    G=np.stack([synth.colmajor16(Tinv@synth._se3(synth._random_rot(rng,np.deg2rad(5)),0.005*rng.standard_normal(3))) for _ in range(n)])
    for split in ("0","1"):
        os.environ["PGP_ICP_SPLIT"]=split
        sc.icp_refine(S,M,G,trim=0.9,max_iterations=10)
        t0=time.perf_counter()
        for _ in range(3): Tr,e,it=sc.icp_refine(S,M,G,trim=0.9,max_iterations=10)
        dt=(time.perf_counter()-t0)/3
        print(f"poses {n:4d} split={split}: {dt*1e3:8.2f} ms/call, {it.sum()/dt:10.0f} pose-iters/s")
