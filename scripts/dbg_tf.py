import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import trace_oracle as orc
from oracle.gen_golden import case_inputs
from trace_amd.sampler import TraceSampler
from trace_amd.weights import random_state_dict
name = sys.argv[1]
c = case_inputs(name); cfg = c['cfg']
s = TraceSampler(cfg, random_state_dict(cfg, 0), 'cuda:0')
s.set_guidance(c['guidance'], c['batch']); s.update_guidance(global_t=c['step_index'])
gold = dict(np.load('tests/golden/sample_%s.npz' % name))
d = torch.tensor(gold['diffusion_steps']); B,N,T = d.shape[:3]
dsc = orc.scale_traj(cfg, d.reshape(B*N,-1,6)).reshape(B*N,T,101,6)
for k in range(100):
    t = 99-k
    out = s.forward(c['batch'], num_samp=N, class_free_guide_w=c['w'], apply_guidance=c['apply'], guide_clean=c['clean'],
                    x_init=dsc[:,:,k], steps=[t], noise_steps=c['noise_steps'][k:k+1])
    got = out['pred_traj_scaled'].reshape(B*N,T,6).cpu()
    ref = dsc[:,:,k+1]
    rel = ((got-ref).abs()/(1+ref.abs()))
    print(k, t, 'worst %.2e' % rel.max().item(), 'per-traj', ['%.1e'%v for v in rel.amax(dim=(1,2)).tolist()])
