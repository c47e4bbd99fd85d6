import torch, torch.nn.functional as F, sys
sys.path.insert(0,'.')
from advchain_amd import ops
from tests.helpers import rand
def to_planar(g):
    d=g.shape[-1]; return g.permute(0,d+1,*range(1,d+1)).contiguous()
C=4; dims=(8,12,16)
inp=rand((2,C)+dims,1); grid=rand((2,)+dims+(3,),2,-1.3,1.3)
ref=F.grid_sample(inp,grid,mode='bilinear',padding_mode='zeros',align_corners=True)
out=ops.raw_grid_sample_fwd(inp.cuda(),to_planar(grid).cuda(),0,0,False).cpu()
torch.set_printoptions(precision=4, linewidth=200)
for c in range(4):
    print('c',c,'out',out[0,c,0,0]); print('    ref',ref[0,c,0,0])
# search: where does out[0,0,0,0,4] appear in ref?
v=out[0,0,0,0,4]
print('match', ((ref-v).abs()<1e-6).nonzero().tolist())
v=out[0,0,0,0,5]
print('match', ((ref-v).abs()<1e-6).nonzero().tolist())
