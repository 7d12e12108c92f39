import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from importlib import import_module
from oracle import ops_np as O
from util_params import shuffle_fn, randomise, export
model=import_module('efficientlo-net_amd.model'); perm_mod=import_module('efficientlo-net_amd.perm'); synth=import_module('efficientlo-net_amd.synth'); mu=import_module('efficientlo-net_amd.model_util')
DEV='cuda:0'; t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
H,W,B=64,900,1
f1,f2=synth.frame_pair(B,H,W,seed=12)
pad=np.zeros((B,1000,3),np.float32)
cloud=np.concatenate([np.concatenate([f1.reshape(B,-1,3),pad],1),np.concatenate([f2.reshape(B,-1,3),pad],1)],1)
cloud6=np.concatenate([cloud,np.zeros_like(cloud)],-1)
eye=np.tile(np.eye(4,dtype=np.float32),(B,1,1))
net=model.PWCLONet(DEV,seed=2,perm_source=perm_mod.PermSource(fn=shuffle_fn))
out=net.forward_points(t(cloud6),H,W,t(eye),t(eye),t(eye),aug_frame=np.array([1]))
randomise(net.store,seed=4)
out=net.forward_points(t(cloud6),H,W,t(eye),t(eye),t(eye),aug_frame=np.array([1]))
n=cloud.shape[1]//2
p1,p2,q_gt,t_gt=O.PreProcess(cloud[:,:n],cloud[:,n:],eye,eye,eye,np.array([1]))
g=mu.PreProcess(t(cloud[:,:n]),t(cloud[:,n:]),t(eye),t(eye),t(eye),np.array([1]))
print("preprocess diff", np.abs(g[0].cpu().numpy()-p1).max(), np.abs(g[1].cpu().numpy()-p2).max())
xa=[mu.ProjectPC2SphericalRing(t(p1),None,H,W)[0] for _ in range(3)]
print("projection repeatable", torch.equal(xa[0],xa[1]), torch.equal(xa[1],xa[2]), (xa[0]!=xa[1]).sum().item())
x1=xa[0].cpu().numpy(); x2=mu.ProjectPC2SphericalRing(t(p2),None,H,W)[0].cpu().numpy()
o2=net.forward(t(x1),t(x2))
print("fwd_points vs forward(x1,x2)", [float((a-b).abs().max()) for a,b in zip(out[:8],o2[:8])])
want=O.get_model_from_projection(export(net.store),shuffle_fn,x1,x2)
print("forward(x1,x2) vs oracle", [float(np.abs(a.cpu().numpy()-b).max()) for a,b in zip(o2[:8],want[:8])])
o3=net.forward(t(x1),t(x2))
print("forward repeatable", [float((a-b).abs().max()) for a,b in zip(o3[:8],o2[:8])])
pc=t(cloud6)
with torch.no_grad():
    a1,a2,_,_=mu.PreProcess(pc[:, :n, 0:3], pc[:, n:, 0:3], t(eye),t(eye),t(eye), np.array([1]))
    print("pre (sliced input) vs oracle", float((a1.cpu()-torch.from_numpy(p1)).abs().max()), a1.is_contiguous(), a1.stride())
    y1=mu.ProjectPC2SphericalRing(a1,None,H,W)[0]
    print("proj internal vs x1", float((y1.cpu()-torch.from_numpy(x1)).abs().max()), int((y1.cpu()!=torch.from_numpy(x1)).sum()))
    b1=t(p1)
    print("bit diffs", int((a1.view(torch.int32)!=b1.view(torch.int32)).sum()))
    ya=mu.ProjectPC2SphericalRing(a1.clone(),None,H,W)[0]; yb=mu.ProjectPC2SphericalRing(b1,None,H,W)[0]; yc=mu.ProjectPC2SphericalRing(b1.clone(),None,H,W)[0]
    print("a1 vs a1.clone", int((y1!=ya).sum()), " b1 vs b1.clone", int((yb!=yc).sum()), " a1 vs b1", int((y1!=yb).sum()))
    d=(a1.view(torch.int32)!=b1.view(torch.int32)).nonzero()[:5]
    print(d, a1[d[:,0],d[:,1],d[:,2]], b1[d[:,0],d[:,1],d[:,2]])
