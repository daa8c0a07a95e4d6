import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import load_protein
from ai2bmd_amd.fragmentation import build_plan, fragment_positions
from ai2bmd_amd.visnet_calculator import ViSNetEngine
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
dev="cuda:0"; hp=default_hparams(); sd=make_state_dict(hp, seed=2024)
NF=int(sys.argv[1]) if len(sys.argv)>1 else 4096
LANES=int(sys.argv[2]) if len(sys.argv)>2 else 2
engs=[ViSNetEngine(hp, sd, dev) for _ in range(LANES)]
rng=np.random.default_rng(1234); pool=[]
for pname in ("chig","trpcage","ww","abd"):
    pr=load_protein(pname); pl=build_plan(pr); fp=fragment_positions(pl, pr.positions)
    for b in range(len(pl.start)): pool.append((pl.z[pl.start[b]:pl.end[b]], fp[pl.start[b]:pl.end[b]]))
parts=[]
per=NF//LANES
for l in range(LANES):
    zs,ps,sizes=[],[],[]
    for i in range(l*per,(l+1)*per):
        zf,pf=pool[i%len(pool)]; zs.append(zf); ps.append(pf-pf.mean(0)+rng.normal(0,0.05,size=pf.shape)); sizes.append(len(zf))
    end=np.cumsum(sizes); start=end-np.asarray(sizes)
    z=torch.as_tensor(np.concatenate(zs),dtype=torch.int64).to(dev); pos=torch.as_tensor(np.concatenate(ps),dtype=torch.float32).to(dev)
    parts.append((z,pos,start,end,torch.empty(len(start),device=dev),torch.empty(len(z),3,device=dev)))
streams=[torch.cuda.Stream(device=dev) for _ in range(LANES)]
def step():
    for l in range(LANES):
        z,pos,start,end,e,f=parts[l]
        engs[l].forces_device(z,pos,start,end,e,f,stream=streams[l])
for _ in range(2): step()
torch.cuda.synchronize()
t0=time.perf_counter(); K=4
for _ in range(K): step()
torch.cuda.synchronize()
el=time.perf_counter()-t0
print(f"lanes {LANES} frags {NF}: {K*per*LANES/el:.0f} fragments/s  ({1e3*el/K:.1f} ms/step)")
