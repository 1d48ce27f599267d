"""Does the dynamics launch overlap with the pair sweep when issued on a second stream?"""
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N,d,n,R,M=64,2,10,0,8
Y=synth.swarm_control_points(N,d,n); B=N*d*(n-1)+1
polys=synth.polygon_obstacles(M); ppts,poff=synth.pack_polys(polys); pa,pb=synth.swarm_pairs(N,M)
dev=torch.device('cuda'); f64=torch.float64
s1=torch.cuda.Stream(); s2=torch.cuda.Stream()
cA=_capi.Context(N,d,n,R); cB=_capi.Context(N,d,n,R)
cA.set_polygons(ppts,poff); cA.set_hull_pairs(pa,pb)
d0=torch.from_numpy(Y).to(dev); dY=torch.empty((B,N*d,n+1),dtype=f64,device=dev)
cA.set_stream(torch.cuda.current_stream().cuda_stream)
cA.fd_batch_dev(d0.data_ptr(),1,1.49e-8,B,dY.data_ptr()); torch.cuda.synchronize()
d_tf=torch.full((B,),10.0,dtype=f64,device=dev)
P=cA.num_pairs; L=21; Ps=len(pa)
o_sep=torch.empty((B,P*L),dtype=f64,device=dev); o_sp=torch.empty((B,N*L),dtype=f64,device=dev); o_an=torch.empty((B,N*41),dtype=f64,device=dev)
g_flag=torch.empty((B,Ps),dtype=torch.int32,device=dev); g_p1=torch.empty((B,Ps,3),dtype=f64,device=dev); g_p2=torch.empty((B,Ps,3),dtype=f64,device=dev); g_dist=torch.empty((B,Ps),dtype=f64,device=dev)
def gjk(c): c.pair_sweep_dev(dY.data_ptr(),B,0.9,o_sep.data_ptr(),g_flag.data_ptr(),g_p1.data_ptr(),g_p2.data_ptr(),g_dist.data_ptr(),None,None,128,256)
def bern(c):
    c.dynamics_dev(dY.data_ptr(),d_tf.data_ptr(),B,5.0,True,1.0,o_sp.data_ptr(),o_an.data_ptr())
def run(two,K=300):
    cA.set_stream(s1.cuda_stream); cB.set_stream(s2.cuda_stream if two else s1.cuda_stream)
    for _ in range(20): bern(cB); gjk(cA)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(K): bern(cB); gjk(cA)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/K*1e3
for m in (False,True,False,True): print('two streams' if m else 'one stream ', '%.4f ms/step'%run(m))
