import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from chap_amd.networks import DualDecoder
from oracle import init as oinit
g = np.load('tests/golden/dualdecoder2d_64.npz')
def l2(a,b):
    a=a.detach().double().cpu().numpy() if torch.is_tensor(a) else a.astype(np.float64); b=b.astype(np.float64)
    return np.linalg.norm(a-b)/ (np.linalg.norm(b)+1e-30), np.abs(a-b).max()/(np.abs(b).max()+1e-30)
for mode in ('eval','train'):
  for dt in (torch.float32, torch.bfloat16):
    m = DualDecoder(1,4,{"decoder_type":"mcnet"}).cuda(); m.load_state_dict(oinit.dual_decoder_2d_state(101)); m.set_compute_dtype(dt)
    m.train(mode=='train')
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    kw={}
    if mode=='train':
        masks = oinit.drop_masks_2d(21,2,64,64); kw['drop_masks']={k:v.permute(0,2,3,1).unsqueeze(1).contiguous().cuda() for k,v in masks.items()}
    o = m(x, **kw)
    gen = torch.Generator().manual_seed(11); cots=[torch.randn(t.shape,generator=gen).cuda() for t in o]
    torch.autograd.backward(o,cots)
    pre = 'eval_' if mode=='eval' else 'train64_'
    print(mode, dt, 'logits0', l2(o[0], g[pre+'logits0']), 'dx', l2(x.grad, g[pre+'dx']))
    gr = dict(m.named_parameters())
    for i,n in enumerate(g['grad_pick_names']):
        print('   ', n, l2(gr[str(n)].grad, g[pre+'grad_pick%d'%i]))
