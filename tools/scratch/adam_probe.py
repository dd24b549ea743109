import sys, torch
sys.path.insert(0, '.')
from connectome_gnn_amd.optim import Adam
torch.manual_seed(0)
shapes = [(64, 5), (64,), (300, 7)]
ps_a = [torch.randn(*s, device='cuda').requires_grad_(True) for s in shapes]
ps_b = [p.detach().clone().requires_grad_(True) for p in ps_a]
oa = Adam(ps_a, lr=1e-2, weight_decay=0.0)
ob = torch.optim.Adam(ps_b, lr=1e-2, weight_decay=0.0)
for it in range(4):
    gs = [torch.randn_like(p) * (1.0 + it) for p in ps_a]
    for p, q, g in zip(ps_a, ps_b, gs):
        p.grad, q.grad = g.clone(), g.clone()
    oa.step(); ob.step()
    torch.cuda.synchronize()
    print(it, 'step', float(oa.state[ps_a[0]]['step']), [float((p - q).abs().max()) for p, q in zip(ps_a, ps_b)],
          [float((oa.state[p]['exp_avg'] - ob.state[q]['exp_avg']).abs().max()) for p, q in zip(ps_a, ps_b)])
print('--- round trip')
sd = oa.state_dict()
print('oa sd step', sd['state'][0]['step'], sd['param_groups'][0].get('capturable'))
oa.load_state_dict(ob.state_dict())
print('after load: step obj', oa.state[ps_a[0]]['step'], oa.param_groups[0]['lr'], oa.param_groups[0].get('capturable'))
print('exp_avg equal', [float((oa.state[p]['exp_avg'] - ob.state[q]['exp_avg']).abs().max()) for p, q in zip(ps_a, ps_b)])
for it in range(4, 6):
    gs = [torch.randn_like(p) * (1.0 + it) for p in ps_a]
    for p, q, g in zip(ps_a, ps_b, gs):
        p.grad, q.grad = g.clone(), g.clone()
    oa.step(); ob.step()
    torch.cuda.synchronize()
    print(it, 'step', float(oa.state[ps_a[0]]['step']), float(ob.state[ps_b[0]]['step']), [float((p - q).abs().max()) for p, q in zip(ps_a, ps_b)])
